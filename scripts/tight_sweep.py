"""Development aid (gpurun): the last tile stage's second verdict (csrc/interval_gen.hpp: tight code) under MPR_CTX_PARANOID — every
frame that took a shortcut is rendered a second time the reference's way ON THE CHIP and heights and normals are compared there.

 1. bear (the benchmark's model with sin / cos) under random views — rotated, sheared, mirrored, zoomed, perspective — at 256^3 ..
    2048^3;
 2. random shapes made of what makes a sin / cos matter: primitives under position-dependent rotations (twists, swirls whose angle
    decays with the distance, as bear's does), ripples, hard and smooth blends — small enough for the generated walks (24 slots,
    64 min / max clauses) — under the benchmark's view and random ones at 256^3 / 512^3.
Counts the frames whose last stage ran tight code and the tiles it kept out of the float pass.
usage: tight_sweep.py [VIEWS_OF_BEAR] [FIRST_SEED COUNT]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpr_amd as mpr

NV = int(sys.argv[1]) if len(sys.argv) > 1 else 100
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 0
COUNT = int(sys.argv[3]) if len(sys.argv) > 3 else 300


def view3(rng):
    V = np.eye(4, dtype=np.float32)
    if rng.random() < 0.8:
        V[:3, :3] += rng.uniform(-0.35, 0.35, (3, 3)).astype(np.float32)
    V[:3, :3] *= np.float32(rng.choice([0.6, 1.0, 1.0, 1.7]))
    if rng.random() < 0.3:
        V[int(rng.integers(0, 3))] *= np.float32(-1.0)
    V[:3, 3] = rng.uniform(-0.3, 0.3, 3).astype(np.float32)
    V[3, :3] = rng.uniform(-0.3, 0.3, 3).astype(np.float32)
    return V


def shape(rng):
    X, Y, Z = mpr.Tree.X(), mpr.Tree.Y(), mpr.Tree.Z()
    u = lambda a, b: float(np.float32(rng.uniform(a, b)))
    ax = [X, Y, Z]
    rng.shuffle(ax)
    a, b, c = ax
    # a position-dependent rotation of (a, b)
    kind = rng.integers(0, 4)
    if kind == 0:
        ang = c * u(-4, 4)                                               # twist
    elif kind == 1:
        ang = mpr.exp(-mpr.sqrt(a * a + b * b + c * c) * u(0.5, 2)) * u(-3, 3)   # swirl, decaying
    elif kind == 2:
        ang = mpr.sqrt(a * a + b * b) * u(-5, 5)                         # spiral
    else:
        ang = None
    if ang is not None:
        ca, sa = mpr.cos(ang), mpr.sin(ang)
        a, b = a * ca - b * sa, a * sa + b * ca
    prim = rng.integers(0, 4)
    if prim == 0:
        d = mpr.tmax(mpr.tmax(mpr.tabs(a) - u(0.2, 0.7), mpr.tabs(b) - u(0.1, 0.4)), mpr.tabs(c) - u(0.2, 0.7))
    elif prim == 1:
        q = mpr.sqrt(a * a + b * b) - u(0.3, 0.6)
        d = mpr.sqrt(q * q + c * c) - u(0.08, 0.25)
    elif prim == 2:
        d = mpr.sqrt(a * a * u(0.5, 2) + b * b * u(1, 4) + c * c) - u(0.3, 0.7)
    else:
        d = mpr.tmax(mpr.sqrt(a * a + b * b) - u(0.2, 0.5), mpr.tabs(c) - u(0.3, 0.8))
    if rng.random() < 0.6:                                               # a ripple
        f = u(4, 30)
        d = d + mpr.sin(X * f) * mpr.cos(Y * f * u(0.5, 1.5)) * u(0.01, 0.08)
    other = mpr.sqrt((X - u(-0.5, 0.5)) * (X - u(-0.5, 0.5)) + (Y - u(-0.5, 0.5)) * (Y - u(-0.5, 0.5)) + (Z - u(-0.5, 0.5)) * (Z - u(-0.5, 0.5))) - u(0.1, 0.4)
    how = rng.integers(0, 4)
    if how == 0:
        d = mpr.tmin(d, other)
    elif how == 1:
        d = mpr.tmax(d, -other)
    elif how == 2:
        k = u(8, 40)
        d = mpr.log(mpr.exp(d * -k) + mpr.exp(other * -k)) / -k
    return d


t0 = time.time()
T = np.eye(4, dtype=np.float32)
T[3, 2] = 0.3
grand = [0, 0, 0]
tape = mpr.Tape(mpr.model("bear"))
for S in (256, 512, 1024, 2048):
    n = NV if S <= 1024 else max(NV // 8, 4)
    rng = np.random.default_rng(S * 17 + 3)
    ctx = mpr.Context(S, flags=mpr.CTX_PARANOID)
    plain = mpr.Context(S)
    tight = walked = 0
    for k in range(n):
        V = view3(rng) if k else T
        for _ in range(2):
            ctx.render3D(tape, V)
        if k % 10 == 0:
            plain.render3D(tape, V)
            plain.render3D(tape, V)
            tight += "+tight" in plain.tile_stage_forms()
            walked += plain.frame_tiles()[2]
    st = ctx.paranoid_stats()
    ctx.close()
    plain.close()
    grand = [x + y for x, y in zip(grand, st)]
    print("bear %4d: %3d views x 2 frames, %4d rendered twice, %d cells differ; %d of %d sampled frames ran tight code, %d tiles walked in them; %.0f s"
          % (S, n, st[1], st[2], tight, (n + 9) // 10, walked, time.time() - t0), flush=True)

ctxs = {S: mpr.Context(S, flags=mpr.CTX_PARANOID) for S in (256, 512)}
plain = {S: mpr.Context(S) for S in (256, 512)}
shapes = tight = generated = bad = 0
for seed in range(FIRST, FIRST + COUNT):
    rng = np.random.default_rng(77000 + seed)
    tape = mpr.Tape(shape(rng))
    S = int(rng.choice([256, 512]))
    ctx = ctxs[S]
    before = ctx.paranoid_stats()
    for V in (T, view3(rng)):
        for _ in range(2):
            ctx.render3D(tape, V)
    plain[S].render3D(tape, T)
    plain[S].render3D(tape, T)
    f = plain[S].tile_stage_forms()
    shapes += 1
    tight += "+tight" in f
    generated += "gen" in f
    after = ctx.paranoid_stats()
    if after[2] != before[2]:
        bad += 1
        print("DIFFERS seed %d S %d: %s -> %s (%s)" % (seed, S, before, after, f), flush=True)
for c in ctxs.values():
    grand = [x + y for x, y in zip(grand, c.paranoid_stats())]
print("random shapes, seeds %d..%d: %d shapes x 2 views x 2 frames, %d on generated walks, %d with tight code in the benchmark's view, %d shapes differ; %.0f s"
      % (FIRST, FIRST + COUNT - 1, shapes, generated, tight, bad, time.time() - t0), flush=True)
print("total: %d frames, %d rendered twice, %d cells differ" % tuple(grand))
