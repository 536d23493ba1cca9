"""Dev experiment: how many triangle (x,y)-footprints does a ray's 2-D projection cross before its closest hit
(reference triangle semantics: |t| plane distance, 2-row containment)?  Statistics only, approximate arithmetic."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_raytracing_amd as rtx
from rust_raytracing_amd import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
box = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
objs = scenes.random_triangles(n, seed=2, box=box)
g = objs["geom"].reshape(-1, 3, 3)
v0, v1, v2 = g[:, 0], g[:, 1], g[:, 2]
nrm = np.cross(v1 - v0, v2 - v0)
nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
keep = (nrm * v0).sum(1) >= -1.0
print("triangles kept after n.v0 >= -1:", keep.sum(), "of", n)
v0, v1, v2, nrm = v0[keep], v1[keep], v2[keep], nrm[keep]
lo = np.minimum(np.minimum(v0, v1), v2)[:, :2]
hi = np.maximum(np.maximum(v0, v1), v2)[:, :2]
rng = np.random.default_rng(0)

def tri_dist(p, d):
    cull = (nrm * (v0 - d)).sum(1) < 0
    dn = nrm @ d
    t = np.abs((nrm * (v0 - p)).sum(1) / dn)
    q = p + d * t[:, None]
    # 2-row containment in x,y: solve a*e1 + b*e2 = q - v0 on rows x,y
    e1, e2, r = (v1 - v0)[:, :2], (v2 - v0)[:, :2], (q - v0)[:, :2]
    det = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
    a = (r[:, 0] * e2[:, 1] - r[:, 1] * e2[:, 0]) / det
    b = (e1[:, 0] * r[:, 1] - e1[:, 1] * r[:, 0]) / det
    ok = (~cull) & (a >= 0) & (b >= 0) & (a + b <= 1) & (t > 1e-300) & np.isfinite(t)
    return np.where(ok, t, np.inf)

def crossings(p, d, tmax):
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d[:2]
        t0 = (lo - p[:2]) * inv
        t1 = (hi - p[:2]) * inv
    tn = np.maximum(np.minimum(t0, t1).max(1), 0.0)
    tf = np.maximum(t0, t1).min(1)
    hit = tn <= tf
    return hit.sum(), (hit & (tn <= tmax)).sum()

def stats(name, rays):
    th, call, cpr = [], [], []
    for p, d in rays:
        t = tri_dist(p, d)
        tb = t.min()
        a, b = crossings(p, d, tb)
        th.append(tb); call.append(a); cpr.append(b)
    th = np.array(th); call = np.array(call); cpr = np.array(cpr)
    print("%s: rays %d  miss %.2f  t_hit median %.2f mean(finite) %.2f | footprints crossed: all %.0f, before hit %.0f (median %.0f)" % (
        name, len(rays), np.isinf(th).mean(), np.median(th), th[np.isfinite(th)].mean(), call.mean(), cpr.mean(), np.median(cpr)))

R = 300
prim = []
for _ in range(R):
    ax, ay = (rng.random(2) - 0.5) * np.array([np.pi / 2, np.pi / 2 * 1080 / 1920])
    d = np.array([np.cos(ax) * np.cos(ay), -np.sin(ax), np.sin(ay)])   # rough camera basis, stats only
    prim.append((rng.random(3) * 0.1, d / np.linalg.norm(d)))
stats("primary", prim)
sec = []
for _ in range(R):
    p = np.array([10 + 100 * rng.random(), (rng.random() - .5) * 100 * box, (rng.random() - .5) * 100 * box])
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    sec.append((p, d))
stats("interior", sec)
