#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X: Mrays/s (primary rays, whole node) + achieved HBM GB/s of the
Scene::render hot path on the 10k-sphere 1920x1080 64-spp scene (configs[1], "C2"), at 1/2/4/8 GPUs.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = one render of THE frame the metric names: 1920x1080 at 64 rays per pixel in total, whatever N is (STRONG
scaling).  Rank r of N traces the blocks of 8 rows r, r + N, ... (rtx_render_blocks: whole 8x8 ray tiles; scene resident
in HBM before the timed region), then ONE gather (RCCL over xGMI) brings the bands to rank 0 -- inside the timed region.
For N > 1 the renders of the timed loop are asynchronous (no host round trip between render and gather; the receive
buffer and the row permutation are allocated once); the counters of a step are read from one extra render after it.
value = 1920*1080*64 * K / (max-over-ranks time of the K steps).  --weak multiplies rays_per_pixel by N instead
(fixed rays per GPU; labelled as such, not the metric).  --config C3|C4|C5 benches another BASELINE.json config as the
primary workload (C4 = 3840x2160 at 1024 spp is the one named for 8 GPUs).

stdout carries ONE compact JSON line (< 2000 characters: the contract's keys, `roofline` and `cpu_baseline` as objects of a
few scalars, the other legs as rows of numbers); the FULL record -- everything described below -- goes to --detail-out
(default gpurun_out/bench_detail_n<N>.json; the line names it in `detail`).

At N = 1 with the default config the record also carries
  other_configs   C3 full frame (at its named 64 spp) + the band rank 0 of 8 owns of C3, C4 and C5 + the C5 full frame + J1 (a joint
                  tree: 5k spheres + 50k triangles, not a BASELINE config), at a stated spp: rate + roofline each
  lds_sweep       the LDS-staged f32-filter sweep BASELINE.json's configs[1] describes, same frame, same bits
  cpu_baseline    the CPU oracle on this host's cores, on a sub-sample of the SAME 1920x1080 view

roofline (per kernel the rate comes from).  The path is VALU-issue bound, not HBM bound: the tree and the
filter records are served by L2 / LDS; north_star's ">= 40 % of the HBM roofline" therefore does not apply to it (what an
HBM-bound kernel of this path reaches is resolve_kernel's 5.5 TB/s); the HBM figures are reported next to the VALU ones.
ONE scale for every kernel: a lane-instruction = one VALU instruction on one active lane, a packed instruction (v_pk_*)
counts once, as the hardware counters count it.
  bound      "valu"
  peak       256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.64 T lane-instructions/s (MI355X_MICROARCH.md: a wave64 VALU
             instruction issues over 2 cycles of a SIMD-32)
  achieved   USEFUL lane-instructions per second: what the chosen algorithm needs at least for the tests the kernel counted
             itself in this run (box tests, leaf bounds / filters, exact tests; prices in LANE_OPS) / the launch duration
             measured here with hipEvents on the launch stream.   frac = achieved / peak
  issued     {achieved, frac}: lane-instructions the kernel ISSUED (rocprofv3 SQ_THREAD_CYCLES_VALU = sum over VALU
             instructions of their active lanes) on the same scale -- issued.frac >= frac for every kernel
  lane_utilisation = active lanes per VALU instruction / 64;  valu_busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs x launch
             cycles) = the share of time a SIMD's vector pipe is occupied;  valu_cycles_per_instruction = 4 x
             SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU.  Identity: issued.frac = valu_busy x lane_utilisation x 2 /
             valu_cycles_per_instruction -- the peak assumes the 2-cycle issue, this kernel's min/max/cmp/SGPR-operand mix
             occupies the pipe ~4 cycles per instruction (profiles/valu_rate_ubench.txt), so valu_busy is how close the
             kernel is to what its instruction mix can issue
  kernels    per kernel symbol of a launch: average duration and the same figures from its own counter rows
  stages     (the two-stage sphere kernel) stage 1 (primary rays) and stage 2 (bounces) separately: time, segments, useful
             and issued lane-instructions, lanes
  traffic    HBM-side bytes per launch from rocprofv3 FETCH_SIZE / WRITE_SIZE (separate passes; FETCH doubled as the
             guide prescribes for gfx950, raw figure beside it); hbm_gbs = traffic / launch time, hbm_frac = / 8 TB/s
The counters are collected IN THIS RUN when rocprofv3 is on the PATH (child processes `rocprofv3 --pmc ... -- python3
bench.py --pmc-leg ...` after the timed region; --no-pmc skips them); otherwise they are read from profiles/pmc_counters.json
if its kernel-source hash matches this tree, else the fields are null and `counters_source` says so.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12      # 78.64: one wave64 VALU instruction per 2 cycles per SIMD-32
CONFIGS = {
    "C2": dict(scene="spheres", n=10000, seed=1, box=1.0, w=1920, h=1080, spp=64,
               name="10k random spheres (scene seed 1), 1920x1080"),
    "C3": dict(scene="triangles", n=100000, seed=2, box=1.0, w=1920, h=1080, spp=64,
               name="100k random triangles (scene seed 2), 1920x1080"),
    "C4": dict(scene="spheres", n=10000, seed=1, box=1.0, w=3840, h=2160, spp=1024,
               name="10k random spheres (scene seed 1), 3840x2160"),
    "C5": dict(scene="triangles", n=1000000, seed=3, box=2.0, w=3840, h=2160, spp=256,
               name="1M random triangles (scene seed 3, box x2), 3840x2160, flat BVH"),
    # not a BASELINE.json config: Scene.objects is an arbitrary mix (scene.rs:79-85, object.rs:9-15) and a scene with spheres AND a
    # mesh walks a JOINT tree (wf_trace_packet_kernel<0>, trace_bvh_mesh_kernel<.,0,.>): this leg puts a rate and counters on that path
    "J1": dict(scene="joint", n=55000, n_spheres=5000, n_tris=50000, seed=4, box=1.0, w=1920, h=1080, spp=8,
               name="joint tree: 5k random spheres (seed 4) + 50k random triangles (seed 5), 1920x1080"),
}
KERNEL_NAMES = {1: "exact f64 sweep (trace_exact_kernel)", 2: "LDS-staged f32 filter sweep + exact f64 (trace_mixed_kernel)",
                3: "trace_mixed_kernel + verify",
                4: "flat 4-wide BVH (sphere tree: from 2^20 rays per launch on in two stages -- stage 1 trace_sph_packet_kernel: the "
                   "primary rays, one wave-uniform f32 walk per 8x8 tile through the scalar cache, exact f64 tests + ray_hit with all "
                   "lanes; stage 2 trace_bvh_spheres_kernel<.,2,2>: the rays that survived their first hit, from a queue of 64-byte "
                   "records, per-lane walks over 64-byte nodes (node visits and leaf visits apart, rounds cut when few lanes still walk) -- both "
                   "counted into the launch; trace_bvh_kernel when the tree holds triangles)",
                5: "flat 4-wide BVH, regrouping schedule (trace_bvh_mesh_kernel: f32-only traversal step, exact f64 tests in the "
                   "shading phase; trace_bvh_regroup_kernel when the tree holds no triangles)",
                6: "wavefront form (what AUTO runs for a pure mesh): level 0 = wf_generate_kernel, wf_trace_packet_kernel (f32 only, one "
                   "wave-uniform walk per 8x8 tile of primary rays) and wf_shade_kernel (f64: exact tests, ray_hit), ray state in HBM; "
                   "then trace_bvh_mesh_kernel continues from the level-1 queue (or, for an L2-resident tree with >= 2^24 rays per "
                   "launch, wf_trace_kernel + wf_shade_kernel per bounce level); the roofline object covers the whole sequence of "
                   "one launch, wf_trace_packet_kernel is ~55 % of it"}
KERNEL_SHORT = {1: "trace_exact_kernel", 2: "trace_mixed_kernel (LDS sweep)", 3: "trace_mixed_kernel+verify",
                4: "sphere tree: packets + per-lane walks", 5: "trace_bvh_mesh_kernel (regroup)",
                6: "wavefront: packets + trace_bvh_mesh_kernel"}
# substring of the rocprofv3 Kernel_Name rows that belong to a kernel id (RTX_KERNEL_BVH runs one of two kernels)
KERNEL_SYMBOL = {1: ("trace_exact_kernel",), 2: ("trace_mixed_kernel",), 3: ("trace_mixed_kernel",),
                 4: ("trace_bvh_kernel", "trace_bvh_spheres_kernel", "trace_sph_packet_kernel"), 5: ("trace_bvh_regroup_kernel", "trace_bvh_mesh_kernel", "trace_bvh_spheres_pool_kernel"),
                 6: ("wf_trace_packet_kernel", "wf_trace_kernel", "wf_shade_kernel", "wf_generate_kernel", "trace_bvh_mesh_kernel",
                     "wf_trace_spheres_kernel", "wf_shade_spheres_kernel", "wf_generate_spheres_kernel")}
PMC_LEG_RENDERS = 2                # renders of the workload a --pmc-leg child does (pmc_leg)
# Prices of the algorithmic counts in lane-INSTRUCTIONS (what the chosen algorithm needs at least; a packed instruction
# counts once, an f64 instruction once -- the scale of SQ_THREAD_CYCLES_VALU).  box test: 6 fma + 9 min/max + 2 widen + 2 cmp;
# sphere leaf bound (closest-approach terms, rtx_traverse.h): 30; triangle filter: 16; exact sphere test (sphere.rs:19-30):
# 17 f64 add/mul + sqrt + div (~14 f64 instructions each) = 45; exact triangle test (triangle.rs:108-127): ~40 f64 add/mul +
# 3 div = 82; the LDS sweep's filter: 28 v_pk_fma_f32 per (4 spheres x 2 ray slots) + the sign reduction = 4 per (ray, sphere).
LANE_OPS = {"box": 19.0, "sphere_filter": 30.0, "sweep_filter": 4.0, "tri_filter": 16.0, "sphere_exact": 45.0, "tri_exact": 82.0}
PMC_PASSES = (("fetch", ["FETCH_SIZE", "GRBM_GUI_ACTIVE"]), ("write", ["WRITE_SIZE"]),
              ("sq", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
                      "SQ_WAIT_ANY", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_ANY"]))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS), help="BASELINE.json config benched as the primary workload")
    ap.add_argument("--spp", type=int, default=None, help="total rays per pixel (default: the config's own, 64 for C2)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: rays_per_pixel = spp * N (fixed rays per GPU); not the metric")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto (default), 1 exact f64, 2 LDS sweep, 4 BVH lock-step, 5 BVH regroup, 6 wavefront")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lds-sweep", action="store_true", help="skip the secondary measurement of the LDS sweep kernel")
    ap.add_argument("--no-other-configs", action="store_true", help="skip C3 / C4 band / C5 band at N = 1")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect rocprofv3 counters in this run")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU time budget of each cpu_baseline mode")
    ap.add_argument("--other-spp", default="C3=64,C3band=64,C4=64,C5=1,C5band=4,J1=8", help="rays per pixel of the other_configs legs")
    ap.add_argument("--detail-out", default=None,
                    help="file that receives the FULL record (per-kernel counters, every leg's roofline object, the log); default "
                         "gpurun_out/bench_detail_n<N>.json under the repo root.  stdout carries ONE compact line (< 2000 characters)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 rehearsal on a 1-GPU box: every rank renders on cuda:0 and the gather goes through gloo on host "
                         "copies (RCCL refuses two ranks on one device); the line is marked and is not a measurement")
    ap.add_argument("--pmc-leg", default=None, help=argparse.SUPPRESS)      # internal: CONFIG:SPP:BAND:KERNEL, run under rocprofv3
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------------
def make_objects(cfg):
    from rust_raytracing_amd import scenes
    if cfg["scene"] == "spheres":
        return scenes.random_spheres(cfg["n"], cfg["seed"], box=cfg["box"])
    if cfg["scene"] == "joint":                                  # spheres first: scene order decides ties (scene.rs:250)
        import numpy as np
        return np.concatenate([scenes.random_spheres(cfg["n_spheres"], cfg["seed"], box=cfg["box"]),
                               scenes.random_triangles(cfg["n_tris"], cfg["seed"] + 1, box=cfg["box"])])
    return scenes.random_triangles(cfg["n"], cfg["seed"], box=cfg["box"])


class Acc:
    """Sums RtxStats over the timed steps."""

    def __init__(self):
        self.trace_ms = self.stage1_ms = 0.0
        self.segments = self.filter = self.exact = self.box = self.launches = 0
        self.s1_box = self.s1_filter = self.s1_exact = 0
        self.primary = 0
        self.kernel = 0
        self.n = 0

    def add(self, st):
        self.trace_ms += st.trace_ms
        self.stage1_ms += st.stage1_ms
        self.segments += st.segments
        self.filter += st.filter_tests
        self.exact += st.exact_tests
        self.box += st.box_tests
        self.s1_box += st.stage1_box_tests
        self.s1_filter += st.stage1_filter_tests
        self.s1_exact += st.stage1_exact_tests
        self.primary += st.primary_rays
        self.launches += st.trace_launches
        self.kernel = st.kernel
        self.n += 1


def kernel_source_hash():
    """Hash of the kernel sources: a counter file is only trusted for the tree it was measured on."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "rust-raytracing_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def useful_ops(kernel, tri, box, leaf, exact, filt=0.0):
    """(lane-instructions, operand bytes) the algorithm needs at least for these test counts (LANE_OPS)."""
    if kernel in (4, 5, 6):
        return (box * LANE_OPS["box"] + leaf * LANE_OPS["tri_filter" if tri else "sphere_filter"] +
                exact * LANE_OPS["tri_exact" if tri else "sphere_exact"],
                box * 32.0 + leaf * (32.0 if tri else 16.0) + exact * (112.0 if tri else 32.0))
    if kernel in (2, 3):
        return (filt * LANE_OPS["tri_filter" if tri else "sweep_filter"] + exact * LANE_OPS["tri_exact" if tri else "sphere_exact"],
                filt * (32.0 if tri else 16.0) + exact * (112.0 if tri else 32.0))
    return exact * LANE_OPS["tri_exact" if tri else "sphere_exact"], exact * (112.0 if tri else 32.0)


# The work round 3's algorithm did per ray segment (lane-instructions of the tests it counted: profiles/r03_bench_n1.json, roofline.
# algorithmic.lane_ops_per_segment).  roofline.frac counts the tests a kernel PERFORMS, so it falls when a kernel stops doing tests it does
# not need (round 4's tile lists: 131 box tests per primary ray -> none); frac_r03_work = segments x THIS fixed figure / time / peak is the
# same quantity against a fixed amount of work per unit, comparable across rounds.
R03_LANE_OPS_PER_SEGMENT = 2124.098          # C2 / C4 (the 10k-sphere scene)


def algorithmic(acc, cfg):
    """What the kernel counted in this run, priced in lane-instructions and bytes (unit = one ray segment)."""
    steps = max(acc.n, 1)
    seg = acc.segments / steps
    tri = cfg["scene"] in ("triangles", "joint")                 # (a joint tree's leaf tests are priced as triangle tests: 10 of 11 shapes)
    out = {"segments_per_launch": seg / max(acc.launches / steps, 1)}
    if acc.kernel in (4, 5, 6):
        box, leaf, exact = acc.box / steps, (acc.filter - acc.box) / steps, acc.exact / steps
        ops, nbytes = useful_ops(acc.kernel, tri, box, leaf, exact)
        out.update({"box_tests_per_segment": box / seg if seg else 0.0, "leaf_filter_tests_per_segment": leaf / seg if seg else 0.0,
                    "exact_tests_per_segment": exact / seg if seg else 0.0})
    elif acc.kernel in (2, 3):
        filt, exact = acc.filter / steps, acc.exact / steps
        ops, nbytes = useful_ops(acc.kernel, tri, 0.0, 0.0, exact, filt)
        out.update({"filter_tests_per_segment": filt / seg if seg else 0.0, "exact_tests_per_segment": exact / seg if seg else 0.0})
    else:
        exact = acc.exact / steps
        ops, nbytes = useful_ops(acc.kernel, tri, 0.0, 0.0, exact)
    avg_ms = acc.trace_ms / steps
    fixed = R03_LANE_OPS_PER_SEGMENT if (cfg.get("scene") == "spheres" and cfg.get("n") == 10000 and acc.kernel == 4) else None
    if fixed and avg_ms > 0:
        out["frac_r03_work"] = fixed * seg / (avg_ms * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS
    out.update({"lane_ops_per_step": ops, "bytes_per_step": nbytes, "bytes_per_segment": nbytes / seg if seg else 0.0,
                "lane_ops_per_segment": ops / seg if seg else 0.0,
                "Tlane_ops_per_s": ops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0,
                "frac_of_valu_peak": ops / (avg_ms * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS if avg_ms > 0 else 0.0,
                "operand_GBs": nbytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
                "prices_lane_instructions": LANE_OPS,
                "note": "counted by the kernel in this run; operands are served by L2 / LDS, so operand_GBs is not an HBM rate"})
    return out


def sq_figures(c, t):
    """issued lane-instructions/s, lane utilisation, valu_busy ... of one set of counters over `t` seconds."""
    out = {}
    if t > 0 and c.get("SQ_THREAD_CYCLES_VALU") is not None:
        out["issued_Tlane_ops_per_s"] = c["SQ_THREAD_CYCLES_VALU"] / t / 1e12
        out["issued_frac"] = out["issued_Tlane_ops_per_s"] / VALU_PEAK_TLANEOPS
        if c.get("SQ_ACTIVE_INST_VALU"):
            # active lanes per VALU instruction / 64 (SQ_INSTS_VALU counts instructions; SQ_ACTIVE_INST_VALU, ~1 tick per
            # instruction, stands in when a stored counter set lacks it)
            out["lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * (c.get("SQ_INSTS_VALU") or c["SQ_ACTIVE_INST_VALU"]))
            # SIMD-cycles the vector pipe is occupied (the counter ticks in units of 4 cycles) over 1024 SIMDs x launch time
            out["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * t * 2.4e9)
            if c.get("SQ_INSTS_VALU"):
                out["valu_cycles_per_instruction"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]
        if c.get("SQ_INSTS_SALU") is not None:
            out["salu_busy"] = c["SQ_INSTS_SALU"] / (256.0 * t * 2.4e9)        # one scalar pipe per CU, ~1 instruction per cycle
        if c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAIT_ANY") is not None:
            out["wait_any_over_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    return out


def _sphere_stage_of(symbol):
    """1 / 2 for the kernels of the sphere tree's two stages (trace_sph_packet_kernel | trace_bvh_spheres_kernel<SPILL, MODE[, Q3]>
    | trace_sph_pool_kernel<...>), else 0."""
    if symbol.startswith("trace_sph_packet_kernel"):
        return 1
    if symbol.startswith("trace_sph_pool_kernel"):
        return 2
    if symbol.startswith("trace_bvh_spheres_kernel<"):
        args = [a.strip() for a in symbol[symbol.index("<") + 1:symbol.rindex(">")].split(",")]
        return int(args[1]) if len(args) > 1 and args[1] in ("1", "2") else 0
    return 0


def roofline_of(acc, cfg, counters, source, per_kernel=None):
    """The roofline object of one measured kernel; `counters` = {counter: value per launch} or None; per_kernel = the same per
    kernel symbol ({symbol: {counter: per launch, "_ms": average duration, "_calls": rows per launch}})."""
    steps = max(acc.n, 1)
    launches_per_step = max(acc.launches / steps, 1)
    avg_ms = acc.trace_ms / max(acc.launches, 1)                  # average duration of ONE launch (hipEvents, launch stream)
    alg = algorithmic(acc, cfg)
    out = {"bound": "valu", "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s", "kernel": KERNEL_NAMES.get(acc.kernel, str(acc.kernel)),
           "kernel_short": KERNEL_SHORT.get(acc.kernel, str(acc.kernel)), "avg_launch_ms": avg_ms, "launches_per_step": launches_per_step,
           "achieved": alg["Tlane_ops_per_s"], "frac": alg["frac_of_valu_peak"],
           "frac_source": "useful lane-instructions (the tests the kernel counted in this run x LANE_OPS) / launch time / peak",
           "issued": None, "lane_utilisation": None, "valu_busy": None, "valu_cycles_per_instruction": None, "effective_clock_ghz": None,
           "traffic": None, "traffic_raw": None, "hbm_gbs": None, "hbm_frac": None,
           "hbm_note": "this path is VALU-issue bound (tree and records are served by L2 / LDS): north_star's >= 40 % of the HBM "
                       "roofline does not apply to it; hbm_frac is reported, the target that replaces it is valu_busy -> 1 with lanes full",
           "counters_source": source, "algorithmic": alg}
    c = counters or {}
    t = avg_ms * 1e-3
    f = sq_figures(c, t)
    if f:
        out["issued"] = {"achieved": f["issued_Tlane_ops_per_s"], "frac": f["issued_frac"],
                         "source": "SQ_THREAD_CYCLES_VALU / launch time / peak"}
        for k in ("lane_utilisation", "valu_busy", "valu_cycles_per_instruction", "salu_busy", "wait_any_over_wave_cycles"):
            out[k] = f.get(k)
        out["identity"] = "issued.frac = valu_busy x lane_utilisation x 2 / valu_cycles_per_instruction"
        out["valu_instructions_per_launch"] = c.get("SQ_INSTS_VALU")
    if t > 0 and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # rocprofv3 reports KB; MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE counts 128-B requests as 64 B -> doubled
        out["traffic_raw"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        out["traffic"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        out["fetch_bytes_corrected"] = 2.0 * c["FETCH_SIZE"] * 1024.0
        out["write_bytes"] = c["WRITE_SIZE"] * 1024.0
        out["hbm_gbs"] = out["traffic"] / t / 1e9
        out["hbm_frac"] = out["hbm_gbs"] / HBM_PEAK_GBS
    if c.get("GRBM_GUI_ACTIVE") and c.get("_pmc_launch_s"):
        out["effective_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / c["_pmc_launch_s"] / 1e9      # of the profiled pass
    out["counters"] = {k: v for k, v in c.items()} if c else None          # per launch, as rocprofv3 reported them (sizes in KB)
    out["counters_per_kernel"] = per_kernel or None
    # ---- per kernel symbol of the launch: its own duration (under the profiler) and counters
    if per_kernel:
        ks = []
        for sym, kc in sorted(per_kernel.items(), key=lambda kv: -kv[1].get("_ms", 0.0)):
            kt = kc.get("_ms", 0.0) * 1e-3
            rec = {"kernel": sym, "avg_ms_per_launch": kc.get("_ms"), "rows_per_launch": kc.get("_calls")}
            rec.update(sq_figures(kc, kt))
            if kt > 0 and "FETCH_SIZE" in kc and "WRITE_SIZE" in kc:
                rec["traffic"] = (2.0 * kc["FETCH_SIZE"] + kc["WRITE_SIZE"]) * 1024.0
                rec["fetch_bytes_corrected"] = 2.0 * kc["FETCH_SIZE"] * 1024.0
                rec["write_bytes"] = kc["WRITE_SIZE"] * 1024.0
                rec["hbm_frac"] = rec["traffic"] / kt / 1e9 / HBM_PEAK_GBS
            ks.append(rec)
        out["kernels"] = ks
    # ---- the two stages of the sphere kernel, separately (RtxStats.stage1_*: time by hipEvent, the tests stage 1 counted)
    if acc.kernel == 4 and acc.stage1_ms > 0 and cfg["scene"] == "spheres":
        tri = False
        s1_ms = acc.stage1_ms / max(acc.launches, 1)
        s2_ms = avg_ms - s1_ms
        seg1 = acc.primary / steps / launches_per_step                     # every primary ray is exactly one segment of stage 1
        seg2 = acc.segments / steps / launches_per_step - seg1
        stages = []
        for name, ms, seg, box, filt, exact, sym in (
                ("stage 1: primary rays", s1_ms, seg1, acc.s1_box, acc.s1_filter - acc.s1_box, acc.s1_exact, 1),
                ("stage 2: the rays that survived their first hit", s2_ms, seg2, acc.box - acc.s1_box,
                 (acc.filter - acc.box) - (acc.s1_filter - acc.s1_box), acc.exact - acc.s1_exact, 2)):
            box, filt, exact = box / max(acc.launches, 1), filt / max(acc.launches, 1), exact / max(acc.launches, 1)
            ops, _ = useful_ops(4, tri, box, filt, exact)
            rec = {"stage": name, "ms": ms, "segments": seg, "Msegments_per_s": seg / (ms * 1e-3) / 1e6 if ms > 0 else None,
                   "box_tests_per_segment": box / seg if seg else None, "leaf_bounds_per_segment": filt / seg if seg else None,
                   "exact_tests_per_segment": exact / seg if seg else None,
                   "useful_Tlane_ops_per_s": ops / (ms * 1e-3) / 1e12 if ms > 0 else None,
                   "frac": ops / (ms * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS if ms > 0 else None}
            for k in out.get("kernels", []):
                if _sphere_stage_of(k["kernel"]) == sym:
                    rec.update({"kernel": k["kernel"], "issued_frac": k.get("issued_frac"), "lane_utilisation": k.get("lane_utilisation"),
                                "valu_busy": k.get("valu_busy"), "salu_busy": k.get("salu_busy"), "ms_under_profiler": k.get("avg_ms_per_launch"),
                                "traffic": k.get("traffic"), "fetch_bytes_corrected": k.get("fetch_bytes_corrected"), "write_bytes": k.get("write_bytes")})
            stages.append(rec)
        out["stages"] = stages
    return out


def _r(x, n=4):
    return round(x, n) if isinstance(x, float) else x


def _sig(x, n=4):
    """x rounded to n significant digits (None and non-floats pass through): the compact line is read by people and parsers."""
    if not isinstance(x, float) or x != x or x in (float("inf"), float("-inf")):
        return None if isinstance(x, float) else x
    return float("%.*g" % (n, x))


COMPACT_LIMIT = 2000          # characters of the ONE stdout line (records keep only a tail of stdout; r03's 53 KB line went unparsed)
OTHER_COLS = ["Mrays_s", "ms", "spp", "frac", "issued", "lanes", "GB", "band_over_full"]


def compact_line(full, detail_path):
    """The contract line: every key the driver's parser reads, `roofline` and `cpu_baseline` as objects of a few scalars, the
    other legs as rows of numbers (columns named once).  Every value is a rounded copy of one in the full record (`detail`)."""
    r = full["roofline"]
    roof = {"bound": r["bound"], "achieved": _sig(r["achieved"]), "peak": _sig(r["peak"]), "unit": r["unit"], "frac": _sig(r["frac"]),
            "frac_r03_work": _sig((r.get("algorithmic") or {}).get("frac_r03_work")),
            "traffic": _sig(r["traffic"]) if r.get("traffic") else None,
            "issued_frac": _sig((r.get("issued") or {}).get("frac")), "lane_utilisation": _sig(r.get("lane_utilisation"), 3),
            "valu_busy": _sig(r.get("valu_busy"), 3), "hbm_frac": _sig(r.get("hbm_frac"), 3),
            "avg_launch_ms": _sig(r["avg_launch_ms"]), "kernel": r.get("kernel_short"),
            "counters": "live" if "of this run" in (r.get("counters_source") or "") else
                        ("stored" if "pmc_counters.json (" in (r.get("counters_source") or "") else None)}
    if r.get("stages"):
        roof["stages"] = [{"k": (s_.get("kernel") or s_["stage"])[:24], "ms": _sig(s_["ms"]), "frac": _sig(s_["frac"], 3),
                           "issued": _sig(s_.get("issued_frac"), 3), "lanes": _sig(s_.get("lane_utilisation"), 3),
                           "valu_busy": _sig(s_.get("valu_busy"), 3), "salu_busy": _sig(s_.get("salu_busy"), 3),
                           "GB": _sig(s_["traffic"] / 1e9, 3) if s_.get("traffic") else None} for s_ in r["stages"]]
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    out["value"], out["ms_per_step"] = _sig(full["value"], 6), _sig(full["ms_per_step"], 5)
    if "rehearsal" in full:
        out["rehearsal"] = True
    c = full["config"]
    out["config"] = {"workload": c["workload_short"], "width": c["width"], "height": c["height"], "rays_per_pixel": c["rays_per_pixel"],
                     "n_objects": c["n_objects"], "partition": c["partition_short"]}
    out["segments_per_primary_ray"] = _sig(full["segments_per_primary_ray"], 5)
    out["image_mean"] = _sig(full["image_mean"], 9)
    out["roofline"] = roof
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        out["cpu_baseline"] = {"value": _sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                               "sample": cb["sample_short"], "threads_started": cb["threads_started"],
                               "single_thread_Mrays_s": _sig(cb["single_thread_Mrays_s"]), "faithful_Mrays_s": _sig(cb["faithful_Mrays_s"])}
        out["gpu_over_cpu"] = {"tree_walk": _sig(full["speedup_vs_cpu"].get("primary_rays"), 3),
                               "lds_sweep_like_for_like": _sig(full["speedup_vs_cpu"].get("like_for_like_linear_scan"), 3)}
    if "lds_sweep" in full:
        out["lds_sweep_Mrays_s"] = _sig(full["lds_sweep"]["value"])
    if "other_configs" in full:
        out["other_cols"] = OTHER_COLS
        rows = {}
        for o in full["other_configs"]:
            ro = o["roofline"]
            rows[o["leg"]] = [_sig(o["value"]), _sig(o["ms_per_step"]), o["rays_per_pixel"], _sig(ro.get("frac"), 3),
                              _sig((ro.get("issued") or {}).get("frac"), 3), _sig(ro.get("lane_utilisation"), 3),
                              _sig(ro["traffic"] / 1e9, 3) if ro.get("traffic") else None, _sig(o.get("band_rate_over_full_frame_rate"), 3)]
        out["other_configs"] = rows
    if "partition_balance" in full:
        out["partition_8_max_over_mean"] = _sig(full["partition_balance"]["max_over_mean"], 4)
    if "load_imbalance_max_over_mean" in full:
        out["load_imbalance_max_over_mean"] = _sig(full["load_imbalance_max_over_mean"], 4)
    out["detail"] = detail_path
    text = json.dumps(out, separators=(",", ":"))
    if len(text) >= COMPACT_LIMIT:                                  # never let the line outgrow what a record keeps: drop the extras first
        for k in ("image_mean", "segments_per_primary_ray", "gpu_over_cpu", "partition_8_max_over_mean", "lds_sweep_Mrays_s", "other_cols",
                  "other_configs"):
            out.pop(k, None)
            text = json.dumps(out, separators=(",", ":"))
            if len(text) < COMPACT_LIMIT:
                break
    assert len(text) < COMPACT_LIMIT, len(text)
    return text


def write_detail(full, path, n_gpus):
    """The full record goes to a file; returns the path written (relative to the repo root when inside it) or None."""
    cands = [path] if path else [os.path.join(ROOT, "gpurun_out", "bench_detail_n%d.json" % n_gpus),
                                 os.path.join(tempfile.gettempdir(), "rtx_bench_detail_n%d.json" % n_gpus)]
    for p in cands:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(p)), exist_ok=True)
            with open(p, "w") as fh:
                json.dump(full, fh, indent=1)
            return os.path.relpath(p, ROOT) if os.path.abspath(p).startswith(ROOT + os.sep) else p
        except OSError:
            continue
    return None


# ---------------------------------------------------------------------------------------------------------------------
# rocprofv3 counters, collected in this run
# ---------------------------------------------------------------------------------------------------------------------
def _short_symbol(kernel_name):
    """'void rtx::trace_bvh_spheres_kernel<false, 2>(rtx::SceneView const*, ...' -> 'trace_bvh_spheres_kernel<false, 2>'"""
    n = kernel_name.replace("void ", "").split("(")[0]
    return n.split("rtx::")[-1]


def pmc_collect(leg, kernel_symbol, log, launches_per_render=1):
    """Runs `rocprofv3 --pmc ... -- python3 bench.py --pmc-leg LEG` once per counter pass; returns ({counter: per launch},
    {kernel symbol: {counter: per launch, "_ms": its average duration, "_calls": rows per launch}}, source).

    A "launch" is one trace call of the library (one sample batch): one kernel for the megakernels, the two stages of the
    sphere kernel, the whole sequence of generate / walk / shade kernels for the wavefront form -- the counters of every row
    that matches `kernel_symbol` are summed and divided by the number of launches the child made (PMC_LEG_RENDERS renders x
    launches_per_render)."""
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="rtx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    got, per = {}, {}
    try:
        for name, counters in PMC_PASSES:
            out_dir = os.path.join(tmp, name)
            cmd = [exe, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", out_dir, "--",
                                               sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-leg", leg]
            t0 = time.perf_counter()
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            log.append("pmc %s %s: rc %d, %.1f s" % (leg, name, p.returncode, time.perf_counter() - t0))
            if p.returncode != 0:
                return None, None, "rocprofv3 pass '%s' failed: %s" % (name, (p.stderr or p.stdout)[-300:])
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, None, "rocprofv3 pass '%s' wrote no counter file" % name
            acc, dur, kacc, kdur = {}, [], {}, {}
            n_launches = float(PMC_LEG_RENDERS * max(int(round(launches_per_render)), 1))
            for f in files:
                for r in csv.DictReader(open(f)):
                    if not any(sym in r.get("Kernel_Name", "") for sym in kernel_symbol):
                        continue
                    k = r["Counter_Name"]
                    sym = _short_symbol(r["Kernel_Name"])
                    acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
                    kacc.setdefault(sym, {})
                    kacc[sym][k] = kacc[sym].get(k, 0.0) + float(r["Counter_Value"])
                    if "Start_Timestamp" in r and "End_Timestamp" in r and k == counters[0]:
                        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
                        dur.append(d)
                        kdur.setdefault(sym, []).append(d)
            for k in counters:
                if k in acc:
                    got[k] = acc[k] / n_launches
            for sym, kc in kacc.items():
                per.setdefault(sym, {})
                for k, v in kc.items():
                    per[sym][k] = v / n_launches
                if sym in kdur and "_ms" not in per[sym]:
                    per[sym]["_ms"] = sum(kdur[sym]) / n_launches * 1e3
                    per[sym]["_calls"] = len(kdur[sym]) / n_launches
            if name == "fetch" and dur:
                got["_pmc_launch_s"] = sum(dur) / n_launches           # kernel time of one launch (all its kernels)
                got["_pmc_kernels_per_launch"] = len(dur) / n_launches
            if not any(k in acc for k in counters):
                return None, None, "no rows of %s in pass '%s'" % ("/".join(kernel_symbol), name)
    except Exception as e:                                     # noqa: BLE001 -- a profiler problem must not fail the bench
        return None, None, "rocprofv3: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return got, per, "rocprofv3 --pmc passes of this run (separate passes: %s)" % "; ".join(" ".join(c) for _, c in PMC_PASSES)


def stored_counters(key):
    """profiles/pmc_counters.json, trusted only for the kernel sources it was measured on."""
    path = os.path.join(ROOT, "profiles", "pmc_counters.json")
    try:
        d = json.load(open(path))
    except Exception:                                          # noqa: BLE001
        return None, None, "no counters: rocprofv3 was not run here and profiles/pmc_counters.json is absent"
    if d.get("kernel_source_hash") != kernel_source_hash():
        return None, None, "no counters: profiles/pmc_counters.json was measured on other kernel sources (%s)" % d.get("kernel_source_hash")
    rec = d.get("legs", {}).get(key)
    if not rec:
        return None, None, "no counters: profiles/pmc_counters.json has no leg %s" % key
    return rec, d.get("legs_per_kernel", {}).get(key), "profiles/pmc_counters.json (kernel sources %s, %s)" % (d["kernel_source_hash"], d.get("collected", "?"))


def counters_for(leg, kernel_symbol, allow_live, log, launches_per_render=1):
    if allow_live:
        got, per, src = pmc_collect(leg, kernel_symbol, log, launches_per_render)
        if got:
            return got, per, src
        log.append("live counters unavailable: " + src)
    return stored_counters(leg)


def pmc_leg(spec):
    """Child mode (under rocprofv3): the workload of one leg, PMC_LEG_RENDERS renders."""
    import torch
    import rust_raytracing_amd as rtx
    from rust_raytracing_amd import scenes, tiles
    name, spp, band, kernel = spec.split(":")
    cfg = CONFIGS[name]
    world = 8 if band == "band" else 1
    objs = make_objects(cfg)
    hnd = rtx.Scene.from_packed(rtx.Config(rays_per_pixel=int(spp), seed=scenes.RENDER_SEED, kernel=int(kernel)),
                                rtx.Camera(*scenes.CAMERA), objs).upload(0)
    part = tiles.Partition(cfg["h"], 0, world)
    buf = part.alloc_band(cfg["w"], torch.device("cuda", 0))
    for _ in range(PMC_LEG_RENDERS):
        part.render(hnd, cfg["w"], buf)
    torch.cuda.synchronize()
    hnd.close()


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------------------------------
def cpu_threads_available():
    """(threads to use, how that was decided): min(scheduler affinity, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    how = ["os.cpu_count %d" % n]
    try:
        n = len(os.sched_getaffinity(0))
        how.append("sched_getaffinity %d" % n)
    except AttributeError:
        pass
    quota = None
    try:                                                            # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:                                               # noqa: BLE001
        try:                                                        # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:                                           # noqa: BLE001
            pass
    if quota is not None:
        how.append("cgroup cpu quota %.1f" % quota)
        n = max(1, min(n, int(quota + 0.5)))
    else:
        how.append("no cgroup cpu quota readable")
    return n, ", ".join(how), quota


def cpu_baseline(seconds):
    """The oracle on this host's cores, on a sub-sample of the benchmark's OWN view: pixels (x, y) = (kx*i, ky*j) of the
    1920x1080 C2 frame through rtxo_render_pixels -- first on ONE thread, then on the pool (clean mode: no locks), so that
    the line states the parallelism the pool actually delivered --, and whole rows of the same frame in the reference's
    threading (faithful mode: one OS thread per row, a mutex per object; scene.rs:151, object.rs:50).  The samples are sized
    from pilots so that each leg takes about `seconds` (a third of it for the single-thread leg)."""
    import numpy as np
    from oracle import rtx_oracle as oracle
    from rust_raytracing_amd import scenes
    cfg = CONFIGS["C2"]
    w, h = cfg["w"], cfg["h"]
    cores, how, quota = cpu_threads_available()
    objs = make_objects(cfg)
    sc = oracle.make_scene(objs, scenes.CAMERA, rays_per_pixel=1, seed=scenes.RENDER_SEED)

    def grid(n_target):
        k = max(1, int(round((w * h / max(n_target, 1)) ** 0.5)))
        xs, ys = np.meshgrid(np.arange(k // 2, w, k), np.arange(k // 2, h, k))
        return xs.ravel().astype(np.uint32), ys.ravel().astype(np.uint32), k

    def timed(n_target, threads):
        xs, ys, k = grid(n_target)
        t0 = time.perf_counter()
        _, seg = oracle.render_pixels(sc, w, h, xs, ys, n_threads=threads, want_segments=True)
        dt = time.perf_counter() - t0
        return len(xs) / dt / 1e6, int(seg.sum()) / dt / 1e6, dt, len(xs), int(seg.sum()), k

    pilot1 = timed(256, 1)                                             # one thread: pilot, then ~seconds / 3
    single = timed(max(256, min(pilot1[0] * 1e6 * seconds / 3.0, w * h / 4)), 1)
    pilot = timed(64 * cores, cores)                                   # the pool
    clean = timed(min(pilot[0] * 1e6 * seconds, w * h / 4), cores)
    threads_worth = clean[0] / single[0] if single[0] > 0 else float("nan")
    # faithful: whole rows of the same frame, as many as ~seconds allow at ~1/8 of the clean rate (measured ratio round 1)
    n_rows = int(min(max(clean[0] * 1e6 / 8.0 * seconds / w, 1), h // 2))
    stride = h // n_rows
    t0 = time.perf_counter()
    _, fseg = oracle.render(sc, w, h, n_threads=cores, mode=oracle.MODE_FAITHFUL, row_begin=stride // 2, row_stride=stride,
                            want_segments=True)
    fdt = time.perf_counter() - t0
    f_rows = len(range(stride // 2, h, stride))
    faithful = (f_rows * w / fdt / 1e6, int(fseg.sum()) / fdt / 1e6, fdt, f_rows * w)
    return {
        "value": clean[0], "unit": "Mrays/s", "cores": round(threads_worth, 1), "kind": "port",
        "cores_note": "`cores` = the threads' worth the pool delivered = pool rate / single-thread rate, measured here "
                      "(%d threads were started: %s)" % (cores, how),
        "threads_started": cores, "cgroup_cpu_quota": quota,
        "single_thread_Mrays_s": single[0], "single_thread_sample_pixels": single[3],
        "parallel_efficiency": threads_worth / cores if cores else None,
        "sample": "the benchmark's own view: every %dth row and column of the 1920x1080 C2 frame (%d pixels x 1 spp, %.2f "
                  "segments per ray) through the f64 C restatement of the reference CPU path (the Rust crate cannot be built "
                  "here), clean mode: %d threads, no locks; linear scan over the 10^4 spheres per segment as the reference does "
                  "(scene.rs:243-251)" % (clean[5], clean[3], clean[4] / max(clean[3], 1), cores),
        "sample_short": "every %dth row+col of the C2 frame (%d px, 1 spp), C port of the CPU path, %d threads, %.0f s" % (
            clean[5], clean[3], cores, single[2] + clean[2] + fdt),
        "Msegments_s": clean[1], "segments_per_primary_ray": clean[4] / max(clean[3], 1), "seconds": single[2] + clean[2] + fdt,
        "clean_Mrays_s": clean[0], "faithful_Mrays_s": faithful[0], "faithful_Msegments_s": faithful[1],
        "faithful_sample": "%d whole rows of the same frame (%d rays), one OS thread per row + a mutex per object "
                           "(scene.rs:151, object.rs:50)" % (f_rows, faithful[3]),
    }


# ---------------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.pmc_leg:
        return pmc_leg(args.pmc_leg)
    import torch                       # first: one HIP runtime for torch tensors and librtx_hip
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    rehearse = args.rehearse_on_one_gpu and world > 1
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import rust_raytracing_amd as rtx
    from rust_raytracing_amd import scenes, tiles

    cfg = CONFIGS[args.config]
    W, H = cfg["w"], cfg["h"]
    spp_named = args.spp if args.spp is not None else cfg["spp"]
    spp = spp_named * world if args.weak else spp_named           # strong scaling (default): the frame is the same for every N
    log = []

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def run(handle, w, h, rk, wd, n_warm, n_steps, gather):
        """n_steps timed steps of part rk of wd.  Single part / no gather: every step reads its RtxStats (the trace time of
        the roofline is then hipEvents over the timed region).  With a gather: asynchronous renders, one stats render after."""
        part = tiles.Partition(h, rk, wd)
        band = part.alloc_band(w, dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        sync_stats = not gather

        def step():
            st = part.render(handle, w, band, stream=stream, want_stats=sync_stats)
            if not gather:
                return st, band[:part.n_rows]
            if rehearse:
                torch.cuda.synchronize(dev)
                return st, part.gather(band.cpu(), dst=0)
            return st, part.gather(band, dst=0)

        for _ in range(max(n_warm, 1 if gather else 0)):           # (the gather's buffers are allocated by its first call)
            step()
        fence()
        t0 = time.perf_counter()
        acc, full = Acc(), None
        for _ in range(n_steps):
            st, full = step()
            if st is not None:
                acc.add(st)
        fence()
        elapsed = time.perf_counter() - t0
        if not sync_stats and n_steps > 0:                         # the counts of a step (deterministic), outside the timed region
            st = part.render(handle, w, band, stream=stream, want_stats=True)
            for _ in range(n_steps):
                acc.add(st)
        return elapsed, acc, full, part

    objs = make_objects(cfg)
    rcfg = rtx.Config(rays_per_pixel=spp, seed=scenes.RENDER_SEED, kernel=args.kernel)
    handle = rtx.Scene.from_packed(rcfg, rtx.Camera(*scenes.CAMERA), objs).upload(dev_index)   # resident before the timed region
    elapsed, acc, full, part0 = run(handle, W, H, rank, world, args.warmup, args.steps, gather=world > 1)
    per_rank_segments = None
    if world > 1:
        t = torch.tensor([elapsed, float(acc.segments)], dtype=torch.float64, device="cpu" if rehearse else dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        allseg = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allseg, t)
        per_rank_segments = [int(x[1]) // max(args.steps, 1) for x in allseg]     # load balance of the partition
        elapsed = float(tmax[0])
        total_segments = int(tsum[1])
    else:
        total_segments = acc.segments

    single = world == 1 and args.steps > 0
    live_pmc = single and not args.no_pmc
    lds = None
    if single and args.config == "C2" and not args.no_lds_sweep and acc.kernel != 2:
        handle.set_config(rcfg.with_kernel(rtx.RTX_KERNEL_MIXED))
        e2, acc2, full2, _ = run(handle, W, H, 0, 1, 1, 2, gather=False)
        c2, pk2, src2 = counters_for("C2:%d:full:2" % spp, KERNEL_SYMBOL[2], live_pmc, log, acc2.launches / max(acc2.n, 1))
        lds = {"value": W * H * spp * 2 / e2 / 1e6, "unit": "Mrays/s", "ms_per_step": e2 / 2 * 1e3,
               "image_identical_to_value_kernel": bool(torch.equal(full, full2)) if full is not None else None,
               "roofline": roofline_of(acc2, cfg, c2, src2, pk2)}
        handle.set_config(rcfg)
    image_mean = float(full.mean()) if full is not None else float("nan")
    balance = None
    if single and not args.no_other_configs:
        # load balance of the 8-way partition (blocks of 8 rows dealt out round-robin): segments of every band of THIS frame, 4 spp
        handle.set_config(rcfg.with_rays_per_pixel(4))
        segs = []
        for rk in range(8):
            part = tiles.Partition(H, rk, 8)
            band = part.alloc_band(W, dev)
            segs.append(int(part.render(handle, W, band, want_stats=True).segments))
        balance = {"parts": 8, "block_rows": tiles.ROW_BLOCK, "rays_per_pixel": 4, "segments_per_band": segs,
                   "max_over_mean": max(segs) * 8.0 / max(sum(segs), 1)}
        handle.set_config(rcfg)
    handle.close()
    del full

    others = None
    if single and args.config == "C2" and not args.no_other_configs:
        other_spp = dict(kv.split("=") for kv in args.other_spp.split(","))
        others = []
        full_rate = {}
        for name, band in (("C3", False), ("C3", True), ("C4", True), ("C5", False), ("C5", True), ("J1", False)):
            oc = CONFIGS[name]
            s = int(other_spp.get(name + "band" if band and name + "band" in other_spp else name, 4))
            o_objs = objs if oc["scene"] == cfg["scene"] and oc["n"] == cfg["n"] and oc["seed"] == cfg["seed"] else make_objects(oc)
            t_up = time.perf_counter()
            hnd = rtx.Scene.from_packed(rtx.Config(rays_per_pixel=s, seed=scenes.RENDER_SEED), rtx.Camera(*scenes.CAMERA), o_objs).upload(0)
            t_up = time.perf_counter() - t_up
            wd = 8 if band else 1
            e, a, img, part = run(hnd, oc["w"], oc["h"], 0, wd, 1, 2, gather=False)
            hnd.close()
            rays = part.n_rows * oc["w"] * s * 2
            leg = "%s:%d:%s:0" % (name, s, "band" if band else "full")
            cnt, pk, src = counters_for(leg, KERNEL_SYMBOL.get(a.kernel, ("trace_",)), live_pmc, log, a.launches / max(a.n, 1))
            rec = {
                "config": name, "leg": name + ("band" if band else ""), "rays_per_pixel": s, "workload": "%s, %s at %d spp (the config names %d spp; Mrays/s is a rate)" % (
                    oc["name"], "the band rank 0 of 8 owns (%d rows in blocks of %d)" % (part.n_rows, part.block) if band else "full frame", s, oc["spp"]),
                "value": rays / e / 1e6, "unit": "Mrays/s", "ms_per_step": e / 2 * 1e3,
                "Msegments_per_s": a.segments / e / 1e6, "segments_per_primary_ray": a.segments / rays,
                "scene_upload_s": t_up, "image_mean": float(img.mean()), "roofline": roofline_of(a, oc, cnt, src, pk)}
            if not band:
                full_rate[name] = rec["value"]
            elif name in full_rate:
                rec["band_rate_over_full_frame_rate"] = rec["value"] / full_rate[name]
            elif name == "C4":                          # C4 is C2's scene: its band against the value frame (equal spp by default)
                rec["band_rate_over_full_frame_rate"] = rec["value"] / (W * H * spp * args.steps / elapsed / 1e6) if s == spp else None
            others.append(rec)
            del img, o_objs

    if rank == 0:
        steps = max(args.steps, 1)
        rays_per_step = W * H * spp
        value = rays_per_step * args.steps / elapsed / 1e6 if args.steps else 0.0
        cnt, pk, src = (counters_for("%s:%d:full:%d" % (args.config, spp, args.kernel), KERNEL_SYMBOL.get(acc.kernel, ("trace_",)), live_pmc, log,
                                     acc.launches / max(acc.n, 1))
                        if world == 1 else (None, None, "counters are collected at N = 1 only"))
        roof = roofline_of(acc, cfg, cnt, src, pk)
        line = {
            "metric": "Mrays/s (primary rays, whole node), %s %dspp%s" % (
                "10k-sphere 1080p" if args.config in ("C2",) else args.config, spp_named, " per GPU (weak scaling)" if args.weak else ""),
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak" if args.weak else "strong",
            **({"rehearsal": "all ranks on cuda:0, gloo gather through host memory: not a measurement"} if rehearse else {}),
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s, %d rays per pixel in total, max_bounces 10, render seed 42" % (args.config, cfg["name"], spp),
                       "workload_short": "%s: %s, %d spp" % (args.config, cfg["name"], spp),
                       "partition_short": "blocks of %d rows round-robin over %d ranks + 1 gather" % (part0.block, world) if world > 1 else "single GPU",
                       "width": W, "height": H, "rays_per_pixel": spp, "n_objects": cfg["n"],
                       "partition": ("blocks of %d rows dealt out round-robin (rank r renders blocks r, r + %d, ...: whole 8x8 ray tiles), "
                                     "one gather to rank 0 inside the timed region" % (part0.block, world)) if world > 1 else "single GPU",
                       "kernel": KERNEL_NAMES.get(acc.kernel, str(acc.kernel))},
            "Msegments_per_s": total_segments / elapsed / 1e6 if elapsed > 0 else 0.0,
            "segments_per_primary_ray": total_segments / (rays_per_step * steps),
            "image_mean": image_mean,
            "hbm_gbs": roof["hbm_gbs"],
            "roofline": roof,
        }
        if balance is not None:
            line["partition_balance"] = balance
        if per_rank_segments is not None:
            line["segments_per_rank_per_step"] = per_rank_segments
            line["load_imbalance_max_over_mean"] = max(per_rank_segments) * world / max(sum(per_rank_segments), 1)
        if lds is not None:
            line["lds_sweep"] = lds
        if others is not None:
            line["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.cpu_seconds)
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu"] = {"primary_rays": value / cb["value"] if cb["value"] > 0 else None,
                                      "segments": line["Msegments_per_s"] / cb["Msegments_s"] if cb["Msegments_s"] > 0 else None,
                                      "primary_rays_vs_faithful": value / cb["faithful_Mrays_s"] if cb["faithful_Mrays_s"] > 0 else None,
                                      "like_for_like_linear_scan": (lds["value"] / cb["value"] if lds is not None and cb["value"] > 0 else None),
                                      "note": "primary_rays is a tree walk against the CPU's linear scan; like_for_like_linear_scan = the LDS "
                                              "sweep (which scans the list as the reference does) / the same CPU rate"}
        if log:
            line["log"] = log
        detail = write_detail(line, args.detail_out, world)
        sys.stdout.flush()
        print(compact_line(line, detail), flush=True)          # ONE line, < 2000 characters; the full record is in `detail`
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
