"""Builds libnnb variants with other compile-time settings, for timing sweeps on one box:

    python tools/build_variants.py                 # tools/variants/libnnb_<name>.so
    NNB_LIB=tools/variants/libnnb_ts3.so python bench.py --steps 5 --warmup 2 --no-extra-lines --no-cpu-baseline

Every variant has the ABI of the product library (alphazero_amd/_lib.py loads whatever NNB_LIB names)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphazero_amd.build import COMMON, CSRC, SOURCES, _hipcc  # noqa: E402

VARIANTS = {
    "ts3": ["-DNNB_TERMINAL_SIMS_PER_STEP=3"],
    "ts10": ["-DNNB_TERMINAL_SIMS_PER_STEP=10"],
    "ts16": ["-DNNB_TERMINAL_SIMS_PER_STEP=16"],
    "ts32": ["-DNNB_TERMINAL_SIMS_PER_STEP=32"],
    "fcd18": ["-DNNB_FC_SMALL_DEPTH=18"],
    "fcd24": ["-DNNB_FC_SMALL_DEPTH=24"],
    "fcd30": ["-DNNB_FC_SMALL_DEPTH=30"],
    "fcs64": ["-DNNB_FC_SMALL_ROWS=64"],
    "fcs128": ["-DNNB_FC_SMALL_ROWS=128"],
    "noprobe": ["-DNNB_STREAM_PROBE=0"],
    "look32": ["-DNNB_LOOK_ROUNDS=32"],
    "look128": ["-DNNB_LOOK_ROUNDS=128"],
    "look256": ["-DNNB_LOOK_ROUNDS=256"],
    "fc32": ["-DNNB_FC16=0"],
    "fullgrid": ["-DNNB_FC_HALF_GRID=0"],
    "lds3": ["-DNNB_FUSED_LDS_KB=3"],
    "lds5": ["-DNNB_FUSED_LDS_KB=5"],
    "lds7": ["-DNNB_FUSED_LDS_KB=7"],
    "lds10": ["-DNNB_FUSED_LDS_KB=10"],
    "lds13": ["-DNNB_FUSED_LDS_KB=13"],
    "lds19": ["-DNNB_FUSED_LDS_KB=19"],
    "lds26": ["-DNNB_FUSED_LDS_KB=26"],
    "lds31": ["-DNNB_FUSED_LDS_KB=31"],
    "lds52": ["-DNNB_FUSED_LDS_KB=52"],
    "lds78": ["-DNNB_FUSED_LDS_KB=78"],
}
AFFECTED = {"NNB_TERMINAL_SIMS_PER_STEP": ["tree_kernels.hip"], "NNB_LOOK_ROUNDS": ["nnb_api.hip"],
            "NNB_STREAM_PROBE": ["nnb_api.hip"], "NNB_FC_SMALL_ROWS": ["net_kernels_split.hip"],
            "NNB_FC_SMALL_DEPTH": ["net_kernels_split.hip"], "NNB_FUSED_LDS_KB": ["nnb_api.hip"],
            "NNB_FC16": ["net_kernels_split.hip"], "NNB_FC_HALF_GRID": ["net_kernels_split.hip"]}


def main():
    out = os.path.join(ROOT, "tools", "variants")
    os.makedirs(out, exist_ok=True)
    hipcc = _hipcc()
    names = sys.argv[1:] or list(VARIANTS)
    for name in names:
        defs = VARIANTS[name]
        objs = []
        for src, extra in SOURCES:
            base = os.path.join(CSRC, src.replace(".hip", ".o"))
            if any(src in AFFECTED[d[2:].split("=")[0]] for d in defs):
                obj = os.path.join(out, f"{name}_{src.replace('.hip', '.o')}")
                subprocess.check_call([hipcc, *COMMON, *extra, *defs, "-c", os.path.join(CSRC, src), "-o", obj])
            else:
                obj = base                                   # the product build's object
            objs.append(obj)
        lib = os.path.join(out, f"libnnb_{name}.so")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
        print(lib, flush=True)


if __name__ == "__main__":
    main()
