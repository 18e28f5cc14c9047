"""MI355X-native self-play engine behind the nnbattle agent / game API."""
__version__ = "0.1.0"
