"""A numerical study, not part of the product or of the test-suite: how far a Winograd F(2x2, 4x4) form of the two
128->128 convolutions (network.py:31-34) would move their outputs away from the exact result, next to the direct
float32 convolution, on the seed-0 network and 24 random positions.  Run: python oracle/winograd_error_study.py

Result (float32 arithmetic throughout, transformed weights rounded once, error relative to the layer's largest
activation, float64 as the truth):
    direct float32                        layer 2  2.9e-07   layer 3  4.6e-07
    Winograd, points 0, 1, -1, 2, inf     layer 2  1.5e-06   layer 3  2.5e-06
    Winograd, points 0, 1, -1, 1/2, inf   layer 2  1.5e-06   layer 3  1.7e-06
Four to five times the direct form's error: inside the 1e-3 the task allows, outside what keeps network-driven
searches visit-exact against the CPU reference.
"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_port as rp
import torch.nn.functional as F
sd = rp.seeded_state_dict(0)

def mats(points, m, r, dtype=np.float64):
    n = m + r - 1
    def V(cols):
        M = np.zeros((n, cols))
        for j, a in enumerate(points[:-1]):
            M[j] = [a ** k for k in range(cols)]
        M[n - 1] = 0; M[n - 1, cols - 1] = 1
        return M
    Vn, G, Vu = V(n), V(r), V(m)
    BT = np.linalg.inv(Vn).T
    AT = Vu.T
    # balance: scale rows of BT to unit max, G rows inversely
    s = np.abs(BT).max(axis=1)
    BT = BT / s[:, None]; G = G * s[:, None]
    return AT, G, BT

def wino_conv(x, w, AT, G, BT, m=2, r=4, dt=np.float32):
    # x [C, H, W] already zero-padded by 2; w [O, C, 4, 4]; output [O, H-3, W-3]
    C, H, W = x.shape; O = w.shape[0]
    OH, OW = H - 3, W - 3
    n = m + r - 1
    U = np.einsum('ik,ockl,jl->ocij', G, w.astype(np.float64), G).astype(dt)        # transformed weights, rounded once
    th, tw = -(-OH // m), -(-OW // m)
    xp = np.zeros((C, th * m + r - 1, tw * m + r - 1), dt); xp[:, :H, :W] = x
    out = np.zeros((O, th * m, tw * m), dt)
    BTd, ATd = BT.astype(dt), AT.astype(dt)
    for ti in range(th):
        for tj in range(tw):
            d = xp[:, ti * m: ti * m + n, tj * m: tj * m + n]
            Vt = np.einsum('ik,ckl,jl->cij', BTd, d, BTd).astype(dt)
            M = np.einsum('ocij,cij->oij', U, Vt).astype(dt)       # float32 accumulate (numpy pairwise-ish)
            Y = np.einsum('ik,okl,jl->oij', ATd, M, ATd).astype(dt)
            out[:, ti * m:(ti + 1) * m, tj * m:(tj + 1) * m] = Y
    return out[:, :OH, :OW]

rng = np.random.default_rng(0)
pts_sets = {"0,1,-1,2,inf": [0, 1, -1, 2, None], "0,1,-1,1/2,inf": [0, 1, -1, 0.5, None], "0,1,-1,-1/2,inf": [0,1,-1,-0.5,None],
            "0,1/2,-1/2,1,inf": [0, .5, -.5, 1, None], "0,1,-1,2,-2,inf (F(3,4))": None}
boards = []
for _ in range(24):
    b = np.zeros((2, 6, 7), np.float32)
    heights = [0] * 7
    for ply in range(rng.integers(2, 30)):
        c = rng.integers(0, 7)
        if heights[c] < 6:
            b[ply % 2, 5 - heights[c], c] = 1; heights[c] += 1
    boards.append(b)
x0 = torch.tensor(np.stack(boards))
w0, b0 = sd['conv_layers.0.weight'], sd['conv_layers.0.bias']
w2, b2 = sd['conv_layers.2.weight'], sd['conv_layers.2.bias']
w4, b4 = sd['conv_layers.4.weight'], sd['conv_layers.4.bias']
# float64 truth and float32 direct
def chain(dtype):
    a1 = F.relu(F.conv2d(x0.to(dtype), w0.to(dtype), b0.to(dtype), padding=2))
    a2 = F.relu(F.conv2d(a1, w2.to(dtype), b2.to(dtype), padding=2))
    a3 = F.relu(F.conv2d(a2, w4.to(dtype), b4.to(dtype), padding=2))
    return a1, a2, a3
t1, t2, t3 = chain(torch.float64)
d1, d2, d3 = chain(torch.float32)
print("direct f32: layer2 err/max %.3g, layer3 err/max %.3g" % ((d2.double() - t2).abs().max() / t2.abs().max(), (d3.double() - t3).abs().max() / t3.abs().max()))
for name, pts in pts_sets.items():
    if pts is None: continue
    AT, G, BT = mats([p for p in pts], 2, 4)
    # check exactness in float64
    xx = rng.standard_normal((3, 9, 10)); ww = rng.standard_normal((5, 3, 4, 4))
    ref = F.conv2d(torch.tensor(xx)[None], torch.tensor(ww))[0].numpy()
    got = wino_conv(xx, ww, AT, G, BT, dt=np.float64)
    assert np.abs(ref - got).max() < 1e-9, np.abs(ref - got).max()
    e2 = e3 = 0
    for i in range(len(boards)):
        a1 = np.pad(d1[i].numpy(), ((0, 0), (2, 2), (2, 2)))
        y2 = np.maximum(wino_conv(a1, w2.numpy(), AT, G, BT) + b2.numpy()[:, None, None], 0).astype(np.float32)
        e2 = max(e2, np.abs(y2 - t2[i].numpy()).max())
        y3 = np.maximum(wino_conv(np.pad(y2, ((0, 0), (2, 2), (2, 2))), w4.numpy(), AT, G, BT) + b4.numpy()[:, None, None], 0)
        e3 = max(e3, np.abs(y3 - t3[i].numpy()).max())
    print("winograd F(2x2,4x4) f32, points %s: layer2 err/max %.3g, layer3 err/max %.3g" % (name, e2 / t2.abs().max(), e3 / t3.abs().max()))
