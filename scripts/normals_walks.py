"""Development aid: how many tape walks the normals pass makes per 4x4 footprint (one per distinct 16^3 tile among its pixels'
surface voxels) and what a pass that groups pixels by 16^3 tile would make.   python scripts/normals_walks.py bear:1024"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mpr_amd as m
for spec in sys.argv[1:] or ["bear:1024"]:
    name, S = spec.split(":"); S = int(S)
    tape = m.Tape(m.model(name)); T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
    ctx = m.Context(S); ctx.render3D(tape, T)
    h = np.array(ctx.image).reshape(S, S).astype(np.int64); ctx.close()
    filled = h > 0
    pz = np.minimum(h + 1, S - 1)
    zt = np.where(filled, pz >> 4, -1)                       # 16^3 tile layer of the voxel the normal is taken at
    f = zt.reshape(S // 4, 4, S // 4, 4).transpose(0, 2, 1, 3).reshape(S // 4, S // 4, 16)
    fs = np.sort(f, axis=2)
    distinct = (np.diff(fs, axis=2) != 0).sum(axis=2) + 1 - (fs[:, :, 0] == -1)    # distinct layers among filled pixels
    anyf = (f >= 0).any(axis=2)
    distinct = np.where(anyf, distinct, 0)
    walks = int(distinct.sum()); fps = int(anyf.sum()); px = int(filled.sum())
    # regrouped: pixels of one 16^3 tile (16x16 column block x layer) in waves of 16
    yy, xx = np.nonzero(filled)
    key = ((yy >> 4) * (S // 16) + (xx >> 4)) * (S // 16) + zt[yy, xx]
    _, counts = np.unique(key, return_counts=True)
    regrouped = int(((counts + 15) // 16).sum())
    print("%s %d^3: %d pixels, %d footprints with pixels, %d walks (%.2f per footprint; max %d; histogram %s); ideal %d waves of 16 pixels; "
          "grouped by 16^3 tile: %d groups, %d walks" % (name, S, px, fps, walks, walks / max(fps, 1), int(distinct.max()),
          np.bincount(distinct.ravel())[1:9].tolist(), (px + 15) // 16, counts.size, regrouped))
