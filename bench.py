#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X: Mrays/s (primary rays, whole node) of the
Scene::render hot path on the 10k-sphere 1920x1080 scene (configs[1], "C2").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = one full render of the frame: every rank traces its interleaved row band of the image (scene resident
in HBM before the timed region), then ONE gather (RCCL over xGMI) brings the bands to rank 0.  Scaling is weak:
the frame is fixed at 1920x1080 and rays_per_pixel = 64 * N, so every GPU traces the same number of primary
rays (132.7 M) whatever N is.  value = all ranks' primary rays / (max-over-ranks time of the K steps).

Kernels (all produce the same bits; tests/test_gpu_parity.py):
  --kernel 0  AUTO (default) = the flat-BVH kernel for this scene          -> `value`, `roofline`
  --kernel 2  the LDS-staged f32-filter sweep named in BASELINE.json configs[1]; at N=1 it is ALSO timed after the
              main measurement and reported as `lds_sweep` (own roofline), so both designs are on record
  --kernel 1  exact f64 sweep (parity kernel)

roofline (dominant kernel = the trace kernel of rank 0): algorithmic bytes per launch / average launch duration
measured with hipEvents on the launch stream, against the 8 TB/s HBM peak.
  BVH kernel:   bytes = box_tests*32 + leaf_filter_tests*16 + exact_tests*32 + hits*56  (all counted by the kernel)
  LDS sweep:    bytes = segments * n_spheres * 16.  The list is LDS/L2-resident by design, so frac may
                exceed 1: that kernel is VALU-bound, and valu_frac gives the fraction of the VALU issue ceiling.
cpu_baseline: the CPU oracle (a C restatement of the reference's CPU path: kind "port"; the Rust crate cannot be
built here) timed on this host's cores on a bounded sample of the same scene.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch                       # noqa: E402  (first: one HIP runtime for torch tensors and librtx_hip)
import torch.distributed as dist   # noqa: E402

WIDTH, HEIGHT, SPP_PER_GPU, N_SPHERES = 1920, 1080, 64, 10000
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_LANE_OPS_PER_S = 256 * 4 * 32 * 2.4e9      # 256 CU x 4 SIMD-32 x 2.4 GHz
FILTER_OPS_PER_TEST = 8.0          # lane-ops of the f32 sphere filter per (ray, sphere): 7 fma + 1 add
KERNEL_NAMES = {1: "exact f64 sweep (trace_exact_kernel)", 2: "LDS-staged f32 filter sweep + exact f64 (trace_mixed_kernel)",
                3: "trace_mixed_kernel + verify", 4: "flat 4-wide BVH + f32 filter + exact f64 (trace_bvh_kernel)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=None, help="rays per pixel per GPU (default 64: the named config)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto (default), 1 exact f64, 2 LDS sweep, 4 BVH")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lds-sweep", action="store_true", help="skip the secondary measurement of the LDS sweep kernel")
    ap.add_argument("--cpu-sample", default="240x135x1", help="WxHxSPP sample of the same scene for the CPU leg")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N>1 rehearsal on a 1-GPU box: every rank renders on cuda:0 and the gather goes through gloo on host "
                         "copies (RCCL refuses two ranks on one device); the line is marked and is not a measurement")
    return ap.parse_args()


def cpu_baseline(sample):
    """The oracle on this host's cores, bounded sample of the same 10k-sphere scene (rank 0, N=1 only).

    clean mode (thread pool = cores, no locks) renders WxH' with H' = max(H, 4*cores) rows so that every core
    has rows to pull; faithful mode (one OS thread per row + a mutex per object, as scene.rs:151 / object.rs:50)
    renders WxH."""
    from oracle import rtx_oracle as oracle
    from rust_raytracing_amd import scenes
    w, h, spp = (int(v) for v in sample.split("x"))
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    objs = scenes.random_spheres(N_SPHERES, 1)
    sc = oracle.make_scene(objs, scenes.CAMERA, rays_per_pixel=spp, seed=scenes.RENDER_SEED)
    out = {}
    for mode, name, hh in ((oracle.MODE_CLEAN, "clean", max(h, 4 * cores)), (oracle.MODE_FAITHFUL, "faithful", h)):
        t0 = time.perf_counter()
        _, seg = oracle.render(sc, w, hh, n_threads=cores, mode=mode, want_segments=True)
        dt = time.perf_counter() - t0
        out[name] = (w * hh * spp / dt / 1e6, int(seg.sum()) / dt / 1e6, dt, w * hh * spp, hh)
    best = max(out, key=lambda k: out[k][0])
    return {
        "value": out[best][0], "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": "same 10k-sphere scene, %dx%d px x %d spp (%d primary rays), f64 C restatement of the reference "
                  "CPU path (the Rust crate cannot be built here), mode=%s" % (w, out[best][4], spp, out[best][3], best),
        "clean_Mrays_s": out["clean"][0], "faithful_Mrays_s": out["faithful"][0],
        "Msegments_s": out[best][1], "seconds": out["clean"][2] + out["faithful"][2],
    }


class Acc:
    """Sums RtxStats over the timed steps."""

    def __init__(self):
        self.trace_ms = 0.0
        self.segments = self.filter = self.exact = self.box = 0
        self.kernel = 0
        self.n = 0

    def add(self, st):
        self.trace_ms += st.trace_ms
        self.segments += st.segments
        self.filter += st.filter_tests
        self.exact += st.exact_tests
        self.box += st.box_tests
        self.kernel = st.kernel
        self.n += 1


def roofline_of(acc, traffic_key):
    steps = max(acc.n, 1)
    avg_ms = acc.trace_ms / steps
    seg = acc.segments / steps
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": KERNEL_NAMES.get(acc.kernel, str(acc.kernel)),
           "avg_launch_ms": avg_ms, "segments_per_launch": seg}
    if acc.kernel == 4:
        box, leaf, exact = acc.box / steps, (acc.filter - acc.box) / steps, acc.exact / steps
        nbytes = box * 32.0 + leaf * 16.0 + exact * 32.0
        out.update({"algorithmic_bytes_per_segment": nbytes / seg if seg else 0.0,
                    "box_tests_per_segment": box / seg if seg else 0.0, "leaf_filter_tests_per_segment": leaf / seg if seg else 0.0,
                    "exact_tests_per_segment": exact / seg if seg else 0.0,
                    "note": "traversal bytes counted by the kernel (32 B per box test, 16 B per leaf filter record, 32 B per exact "
                            "sphere test); node fetches are dependent L2/HBM reads: latency- and issue-bound, not bandwidth-bound"})
    else:
        nbytes = seg * N_SPHERES * 16.0
        out.update({"algorithmic_bytes_per_segment": N_SPHERES * 16,
                    "valu_frac": (acc.filter / steps * FILTER_OPS_PER_TEST) / (avg_ms * 1e-3) / VALU_LANE_OPS_PER_S
                    if avg_ms > 0 and acc.filter else None,
                    "note": "logical operand bandwidth (list is LDS/L2-resident by design; may exceed the HBM peak)"})
    out["achieved"] = nbytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    out["frac"] = out["achieved"] / HBM_PEAK_GBS
    out["traffic"] = None
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        try:
            rec = json.load(open(tj)).get(traffic_key)
            out["traffic"] = rec["hbm_bytes_per_launch"] if rec else None
        except Exception:
            pass
    # what the memory side actually moved (PMC, profiles/traffic.json) over this run's launch time: the "achieved HBM GB/s"
    # of BASELINE.json's metric.  Far below `achieved`: the operands are served by LDS / L2, the kernels are VALU- and
    # latency-bound
    out["hbm_measured_gbs"] = out["traffic"] / (avg_ms * 1e-3) / 1e9 if out["traffic"] and avg_ms > 0 else None
    return out


def main():
    args = parse()
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

    spp_per_gpu = args.spp if args.spp is not None else SPP_PER_GPU
    spp = spp_per_gpu * world                                  # weak scaling: fixed rays per GPU
    objs = scenes.random_spheres(N_SPHERES, 1)
    cfg = rtx.Config(rays_per_pixel=spp, seed=scenes.RENDER_SEED, kernel=args.kernel)
    scene = rtx.Scene.from_packed(cfg, rtx.Camera(*scenes.CAMERA), objs)
    handle = scene.upload(dev_index)                           # scene resident in HBM before the timed region
    rb, rs, n_rows = tiles.rows_for_rank(HEIGHT, rank, world)
    band = tiles.alloc_band(HEIGHT, WIDTH, world, dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        st = handle.render_rows(WIDTH, HEIGHT, rb, rs, n_rows, band.data_ptr(), stream=stream)
        if rehearse:
            torch.cuda.synchronize(dev)
            full = tiles.gather_bands(band.cpu(), HEIGHT, WIDTH, rank, world, dst=0)
        else:
            full = tiles.gather_bands(band, HEIGHT, WIDTH, rank, world, dst=0)
        return st, full

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(n_warm, n_steps):
        for _ in range(n_warm):
            step()
        fence()
        t0 = time.perf_counter()
        acc, full = Acc(), None
        for _ in range(n_steps):
            st, full = step()
            acc.add(st)
        fence()
        return time.perf_counter() - t0, acc, full

    elapsed, acc, full = timed(args.warmup, args.steps)
    if world > 1:
        t = torch.tensor([elapsed, float(acc.segments)], dtype=torch.float64, device="cpu" if rehearse else dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        total_segments = int(tsum[1])
    else:
        total_segments = acc.segments

    lds = None
    if world == 1 and not args.no_lds_sweep and acc.kernel != 2 and args.steps > 0:
        handle.set_config(cfg.with_kernel(rtx.RTX_KERNEL_MIXED))
        e2, acc2, full2 = timed(1, 2)
        lds = {"value": WIDTH * HEIGHT * spp * 2 / e2 / 1e6, "unit": "Mrays/s", "ms_per_step": e2 / 2 * 1e3,
               "image_identical_to_value_kernel": bool(torch.equal(full, full2)) if full is not None else None,
               "roofline": roofline_of(acc2, "c2_%dspp_kernel2" % spp_per_gpu)}
        handle.set_config(cfg)

    if rank == 0:
        steps = max(args.steps, 1)
        rays_per_step = WIDTH * HEIGHT * spp
        value = rays_per_step * args.steps / elapsed / 1e6 if args.steps else 0.0
        line = {
            "metric": "Mrays/s (primary rays, whole node), 10k-sphere 1080p 64spp",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            **({"rehearsal": "all ranks on cuda:0, gloo gather through host memory: not a measurement"} if rehearse else {}),
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: 10k random spheres (scene seed 1), 1920x1080, %d spp per GPU (%d total), "
                                   "max_bounces 10, render seed 42" % (spp_per_gpu, spp),
                       "width": WIDTH, "height": HEIGHT, "rays_per_pixel": spp, "n_spheres": N_SPHERES,
                       "partition": "interleaved row bands, 1 gather" if world > 1 else "single GPU",
                       "kernel": KERNEL_NAMES.get(acc.kernel, str(acc.kernel))},
            "Msegments_per_s": total_segments / elapsed / 1e6,
            "segments_per_primary_ray": total_segments / (rays_per_step * steps),
            "image_mean": float(full.mean()) if full is not None else float("nan"),
            "roofline": roofline_of(acc, "c2_%dspp_kernel%d" % (spp_per_gpu, acc.kernel)),
        }
        if lds is not None:
            line["lds_sweep"] = lds
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.cpu_sample)
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu"] = value / cb["value"] if cb["value"] > 0 else None
        print(json.dumps(line), flush=True)
    handle.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
