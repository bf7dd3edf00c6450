#!/usr/bin/env python
"""bench.py -- rendered Mtexels/s of the NLT hot path on MI355X (BASELINE.json metric).

One "step" = one full `Model.call` forward (buffers -> two-path U-Net -> pred -> UV->camera warp)
over one batch of synthetic frames already resident in HBM; the batches come out of `Dataset.load_batch`
on a synthetic uint8 capture store and at least three DIFFERENT batches (different frames, different
addresses) are rotated through the timed steps.  Workload at every N: BASELINE
config 3 per GPU -- depth0 16 / depth 256, 4 frames, 1024^2 UV, k = 4 neighbour observation
maps, 512^2 camera-space warp (70 % foreground via fp16, 30 % background = (0,0)); frames shard
data-parallel across ranks with no data-path collective in the forward (weak scaling).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (largest share of the
step's GPU time), timed live with HIP events on the launch stream inside the timed region;
`cpu_baseline` is the CPU oracle (a port: TensorFlow cannot run here) on the host cores, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32-input MFMA peak = the fp32 vector peak (MI355X_MICROARCH.md)
# launch label -> kernel-name fragment in the rocprofv3 output (profiles/*_pmc_traffic.json)
KERNEL_OF_LABEL = {'F.front': ('front4_kernel', 'front_kernel'), 'F.back': ('back_kernel',), 'L0.stem': ('stem_kernel',),
                   'L13.head': ('head_kernel',)}


def pmc_traffic(label):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (FETCH_SIZE / WRITE_SIZE collected in separate --pmc passes, tools/pmc_summary.py applies the
    gfx950 corrections of MI355X_MICROARCH.md).  None when no summary covers this kernel."""
    import glob
    frags = KERNEL_OF_LABEL.get(label, ())
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        for name, rec in d.get('kernels', {}).items():
            if any(fr in name for fr in frags) and rec.get('hbm_bytes') is not None:
                return int(rec['hbm_bytes']), os.path.basename(path)
    return None, None

def live_pmc_traffic(label, args, tune_file):
    """HBM bytes per launch of the dominant kernel MEASURED IN THIS RUN: two child passes of this script's timed forward under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, kernel trace only, as MI355X_MICROARCH.md prescribes;
    tools/pmc_summary.py's reading: FETCH_SIZE doubled for this streaming kernel, WRITE_SIZE as reported), the children loading
    the parent's tile choices so that no plan-time trial launch is in the trace.  (None, reason) when the profiler is not there
    or a pass fails -- the committed figure then stays in the line."""
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if rp is None:
        return None, "rocprofv3 not found"
    if (any(k.startswith(('ROCPROF', 'ROCP_', 'ROCPROFILER')) for k in os.environ)
            or 'rocprofiler' in os.environ.get('LD_PRELOAD', '') or 'rocprofiler' in os.environ.get('HSA_TOOLS_LIB', '')):
        return None, "this process already runs under a profiler (no nested rocprofv3)"
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    try:
        import pmc_summary
    except ImportError as e:
        return None, "tools/pmc_summary.py: %s" % e
    frags = KERNEL_OF_LABEL.get(label, ())
    got = {}
    with tempfile.TemporaryDirectory(dir='/tmp') as td:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(td, counter)
            cmd = [rp, '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', out, '--', sys.executable,
                   os.path.join(ROOT, 'bench.py'), '--headline-only', '--steps', '4', '--warmup', '2', '--precision', args.precision,
                   '--uv', str(args.uv), '--cam', str(args.cam), '--frames', str(args.frames), '--k', str(args.k),
                   '--depth', str(args.depth), '--warp', args.warp, '--tune-cache', tune_file]
            env = dict(os.environ, TMPDIR='/tmp')
            for v in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
                env.pop(v, None)
            try:
                r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=150)
            except (subprocess.TimeoutExpired, OSError) as e:
                return None, "%s pass: %s" % (counter, type(e).__name__)
            if r.returncode != 0:
                return None, "%s pass: rc %d: %s" % (counter, r.returncode, r.stderr.decode(errors='replace')[-160:])
            acc = {name: d for (name, cname), d in pmc_summary.collect(out).items() if cname == counter and d}
            dom = [(len(d), sum(d.values()) / len(d)) for name, d in acc.items() if any(fr in name for fr in frags)]
            if not dom:
                return None, "%s pass: no %s launch in the counter file" % (counter, label)
            calls, avg = max(dom)                                        # (the variant that ran: one kernel name)
            got[counter] = avg * 1024.0                                  # KiB -> bytes
            got[counter + '_step'] = sum(sum(d.values()) for d in acc.values()) * 1024.0 / calls   # every kernel, per forward step
    return {'dominant': int(2 * got['FETCH_SIZE'] + got['WRITE_SIZE']),
            'step': int(2 * got['FETCH_SIZE_step'] + got['WRITE_SIZE_step'])}, None


def pmc_cfg5_bytes(prec):
    """HBM bytes one forward step of BASELINE config 5 (2048^2, 2 frames, k = 1) moves at `prec`, from the newest committed
    tools/pmc_cfg5.sh summary (profiles/*_pmc_traffic_cfg5_<prec>.json)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic_cfg5_%s.json' % prec)), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get('hbm_bytes_per_forward_step'):
            return int(d['hbm_bytes_per_forward_step']), os.path.basename(path)
    return None, None


BYTES_PER_TEXEL = {1: 961.5, 4: 1755.75}   # SURVEY.md 8d, fp32 layer-wise algorithmic bytes
# layer-wise conv FLOP (2 x MAC) per rendered texel of the forward; the obs path adds 4448 per extra neighbour
FLOP_PER_TEXEL = {256: lambda k: 13464 + 4448 * (k - 1), 1024: lambda k: 17944 + 4448 * (k - 1)}


def pmc_step_bytes():
    """HBM bytes one forward step moves, summed over every kernel of the step, from the newest committed PMC summary
    that carries it (tools/pmc_summary.py: sum of calls x (2 x FETCH_SIZE + WRITE_SIZE) / forward steps profiled)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get('hbm_bytes_per_forward_step'):
            return int(d['hbm_bytes_per_forward_step']), os.path.basename(path)
    return None, None


BF16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA (AMD's 5 PF headline includes 2:1 sparsity)


def launch_roof(plan, label):
    """(peak TFLOP/s of fp32-EQUIVALENT work, what it is) for one plan launch: a launch the plan gave to the three-term-split
    kernel (csrc/conv_tile3.hip: `precision = f32x3 / f32x3_9`, the LDS-tiled encoder convs) runs on v_mfma_f32_16x16x32_bf16
    and needs 6 / 9 bf16 products per fp32 product -- its roof is the bf16 peak / NPROD, not the fp32 MFMA peak (review r05,
    weak point 5); everything else is priced against v_mfma_f32_16x16x4_f32."""
    nprod = {'f32x3': 6, 'f32x3_9': 9}.get(plan.precision)
    hint = plan.lds_hints.get(label, 0) & 255
    if nprod and hint and hint != 128 and label not in plan.wino_hints and label not in plan.c32_hints:
        return BF16_MFMA_PEAK_TFLOPS / nprod, "bf16 MFMA peak / %d products" % nprod
    return MFMA_F32_PEAK_TFLOPS, "fp32 MFMA peak"


def whole_pass(args, timer, sec_per_step, layerwise_bpt, live_step_bytes=None, plan=None):
    """Utilisation figures for the whole forward (per GPU): useful FLOP/s against the fp32 MFMA peak AND against the mixed roof
    (every launch priced against the matrix pipe it runs on), and the HBM bytes the step really moves (PMC) against the HBM
    peak -- beside the bytes the plan's launches HAVE to move (compulsory) and what a perfectly fused pass would move.
    SURVEY 8d's layer-wise bytes are what an unfused implementation would move -- kept as a reference figure, NOT as a
    utilisation."""
    flops = sum(timer.flops.get(l, 0) for l in timer.records)            # 2 x MACs of the plan's launches (L0 folded)
    tf = flops / sec_per_step / 1e12
    hbm, src = pmc_step_bytes()
    if live_step_bytes:
        hbm, src = int(live_step_bytes), ("measured in this run (rocprofv3 --pmc child passes; FETCH_SIZE doubled for every kernel: an upper "
                                          "bound for the gather / half-line launches, DESIGN 9)")
    texels = args.frames * args.uv * args.uv
    plan_bytes = int(sum(timer.moved.get(l, r[2]) for l, r in timer.records.items())) + 80 * args.frames * args.cam * args.cam
    fused_bytes = int(texels * (4 * (5 + 3 * args.k) + 12) + 80 * args.frames * args.cam * args.cam)
    out = {"useful_flops_per_step": int(flops), "tflops": round(tf, 2), "frac_of_fp32_mfma_peak": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
           "hbm_bytes_per_step_pmc": hbm, "pmc_source": src,
           "frac_of_hbm_peak": round(hbm / sec_per_step / 1e9 / HBM_PEAK_GBS, 4) if hbm else None,
           "plan_algorithmic_bytes": plan_bytes,
           "plan_algorithmic_bytes_note": "sum over the plan's launches of the bytes each HAS to move (its inputs + outputs once) + the resampler's 80 B per camera pixel",
           "traffic_over_plan_algorithmic": round(hbm / plan_bytes, 3) if hbm else None,
           "perfect_fusion_bytes": fused_bytes, "plan_algorithmic_over_perfect_fusion": round(plan_bytes / fused_bytes, 2),
           "layerwise_equivalent_bytes_per_texel": layerwise_bpt,
           "layerwise_equivalent_GBps": round(texels * layerwise_bpt / sec_per_step / 1e9, 1)}
    if plan is not None:
        # mixed roof: the time the step's FLOPs would take with every launch AT its own matrix-pipe peak, over the step's time
        t_roof, split = 0.0, {}
        for l, r in timer.records.items():
            fl = timer.flops.get(l, 0)
            peak, what = launch_roof(plan, l)
            t_roof += fl / (peak * 1e12)
            if peak != MFMA_F32_PEAK_TFLOPS and fl:
                ms = r[1] / r[0]
                eq = fl / ms / 1e9
                split[l] = {"ms_alone": round(ms, 4), "fp32_equivalent_TFLOPs": round(eq, 1), "frac_of_split_roof": round(eq / peak, 4),
                            "frac_of_fp32_mfma_peak": round(eq / MFMA_F32_PEAK_TFLOPS, 4), "roof": what}
        out["frac_of_mixed_mfma_roof"] = round(t_roof / sec_per_step, 4)
        out["mixed_roof_note"] = ("sum over launches of FLOPs / that launch's matrix-pipe peak (fp32 MFMA 157.3 TF; bf16 MFMA 2500 TF / NPROD for the "
                                  "three-term-split launches), over the step time: the honest matrix-pipe utilisation of a mixed-precision-pipe pass")
        if split:
            out["bf16_mfma_launches"] = split
    return out


def algorithmic_bytes_per_texel(k):
    return 4.0 * (123.8125 + 34.5625 * k + 50.375 + 31.625 * k)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--uv', type=int, default=1024)
    ap.add_argument('--cam', type=int, default=512)
    ap.add_argument('--frames', type=int, default=4, help='frames (light x view pairs) per GPU')
    ap.add_argument('--k', type=int, default=4, help='neighbour observation maps per frame')
    ap.add_argument('--depth', type=int, default=256)
    ap.add_argument('--algo', type=str, default='auto', choices=['auto', 'direct'])
    ap.add_argument('--precision', type=str, default='f32x3_9', choices=['f32x3_9', 'fp32', 'f32x3', 'bf16'],
                    help='f32x3_9 (default; VERDICT r03 ruling): fp32 storage everywhere, the LDS-tiled encoder convs multiply fp32 operands '
                         'as three bf16 terms, all 9 exact term products, fp32 accumulate; fp32: every product on v_mfma_f32 (kept in the line '
                         'as native_fp32); bf16: the middle of the network on bf16 MFMA / bf16 storage (reported as such in dtype)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--graph', action='store_true', help='replay the forward as one hipGraph (model.use_graphs); the dominant '
                    'kernel is then timed in an eager pass of the same steps right after the timed region')
    ap.add_argument('--no-fused', action='store_true', help='layer-by-layer plan (disable csrc/fused.hip) for A/B runs')
    ap.add_argument('--tune-cache', type=str, default=None, help='JSON of tile choices: loaded if present, else written')
    ap.add_argument('--cpu-baseline-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--released-pipelined-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--released-shapes-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--infer-mode-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-threads', type=int, default=0, help='host threads for the CPU oracle leg (0 = all)')
    ap.add_argument('--cpu-frames', type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument('--batches', type=int, default=3, help='distinct batches rotated through the timed steps (>= 3)')
    ap.add_argument('--store-frames', type=int, default=16, help='frames of the synthetic uint8 capture store')
    ap.add_argument('--warp', type=str, default='charts', choices=['charts', 'random'],
                    help="synthetic uv2cam map: 'charts' = piecewise-smooth like a rendered UV pass (default); 'random' = independent "
                         "uniform coordinates per pixel (adversarial: every resampler tap on its own cache line)")
    ap.add_argument('--per-op', action='store_true', help='print a per-launch timing table to stderr')
    ap.add_argument('--dominant', type=str, default=None, help='label of the kernel to time in the timed region')
    ap.add_argument('--train-steps', type=int, default=-1, help='train steps to time per loss for the train_step field (-1: max(50, steps//2), 0: skip)')
    ap.add_argument('--train-loss', type=str, default='l2,barron', help='comma-separated losses, one train_step line each')
    ap.add_argument('--train-graph', action='store_true',
                    help='train step: replay forward + loss + backward as one hipGraph (trainvali.GraphedTrainStep); measured '
                         'slower than eager launches on ROCm 7.0 (5.15 vs 4.75 ms), so off by default')
    ap.add_argument('--headline-only', action='store_true', help='only the timed forward (profiling runs): no loader-inclusive legs, train steps or CPU leg')
    ap.add_argument('--pipelined', action='store_true', help='with --headline-only: also the several-batches-in-flight sub-line')
    ap.add_argument('--no-released-shapes', action='store_true', help='skip the config 1 / config 2 sub-lines (released .ini shapes)')
    ap.add_argument('--per-op-train', action='store_true', help='per-launch timing table of one train step (stderr)')
    ap.add_argument('--no-live-pmc', action='store_true', help='skip the two rocprofv3 --pmc child passes that measure the dominant '
                    "launch's HBM traffic in this run (roofline.traffic then comes from the committed profiles/*_pmc_traffic.json)")
    return ap.parse_args()


def synth_device_batch(n, uv, cam, k, device, seed):
    """A float32 11-tuple of uniform random buffers (tests / tools; the bench itself loads through `make_loader`)."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    U = lambda *s: torch.rand(s, device=device, generator=g)
    base, rgb, cvis, lvis = U(n, uv, uv, 3), U(n, uv, uv, 3), U(n, uv, uv, 1), U(n, uv, uv, 1)
    warp = U(n, cam, cam, 2).half().float()                       # save_float16_npy quantisation
    warp[U(n, cam, cam) >= 0.7] = 0                               # background -> texel (0,0)
    nn_base, nn_rgb = U(n, k, uv, uv, 3), U(n, k, uv, uv, 3)
    rgb_c, nn_rgb_c = U(n, cam, cam, 3), U(n, cam, cam, 3)
    return (None, base, cvis, lvis, warp, rgb, rgb_c, None, nn_base, nn_rgb, nn_rgb_c)


def make_loader(args, device, k, mode, seed, loss='l2'):
    """(config, Dataset on a seeded synthetic uint8 capture store, list of `--batches` disjoint id lists)."""
    import nlt_amd
    from nlt_amd.datasets import get_dataset_class
    from nlt_amd.datasets.synth import synthetic_store
    nb = max(3, args.batches)
    frames = max(args.store_frames, nb * args.frames)
    cfg = nlt_amd.make_config(depth=args.depth, uvh=args.uv, uvw=args.uv, imh=args.cam, imw=args.cam, bs=args.frames,
                              loss=loss, lr=1e-3)
    store = synthetic_store(frames, args.uv, args.cam, device=device, seed=seed, k=k, warp=args.warp)
    ds = get_dataset_class('nlt')(cfg, mode, store, k=k, device=device, ring=nb)
    id_lists = [store['ids'][i * args.frames:(i + 1) * args.frames] for i in range(nb)]
    return cfg, ds, id_lists


def bench_pipelined(args, device, model, batches, lanes_list=(2, 3, 4, 6)):
    """Sub-line: the headline's forward with several batches in flight (nlt_amd.pipeline.RenderPipeline: one lane of render
    state per batch over the same weights; consecutive batches on consecutive lanes).  Same model, same batches, same
    number of steps as the headline; every step's full work is inside the timed region.  `ms_per_step` here is elapsed / steps
    (throughput), not the latency of one batch."""
    import torch
    from nlt_amd.pipeline import RenderPipeline
    out = {"what": "same model, batches and steps as the headline; batch i is queued on lane i % lanes (own plan buffers, launch "
                   "tapes and HIP streams, shared weights), so the launches of up to `lanes` batches overlap on the GPU",
           "lanes": {}}
    ref = model.call(batches[0], 'test')[0].clone()
    single = time_forward(model, batches, max(20, args.steps // 4))
    for lanes in lanes_list:
        pipe = RenderPipeline(model, lanes)
        for _ in range(3):
            for t in [pipe.submit(batches[i % len(batches)], 'test') for i in range(2 * lanes)]:
                t.result()
        same = bool(torch.equal(pipe.submit(batches[0], 'test').result()[0], ref))

        def timed():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tickets = [pipe.submit(batches[i % len(batches)], 'test') for i in range(args.steps)]
            for t in tickets[-lanes:]:
                t.result()
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        dt = timed()
        rec = {}
        if dt / args.steps > 1.15 * single:
            # seen in the first heavy process on a fresh box: every multi-lane leg 1.8-2x slower than one batch at a time while
            # every single-lane leg of the process is normal; gone in any process that follows ~20 s of sustained load (DESIGN.md
            # section 4b).  So: 10 s of the headline's own steps, then measure once more; both figures stay in the line.
            rec["first_try_ms_per_step"] = round(1e3 * dt / args.steps, 4)
            rec["re_measured_after"] = "10 s of sustained single-lane steps"
            t_end = time.perf_counter() + 10.0
            while time.perf_counter() < t_end:
                for i in range(50):
                    model.call(batches[i % len(batches)], 'test')
                torch.cuda.synchronize()
            dt = timed()
        pipe.close()
        del pipe
        rec.update({"ms_per_step": round(1e3 * dt / args.steps, 4),
                    "Mtexels_per_s": round(args.frames * args.uv * args.uv * args.steps / dt / 1e6, 1),
                    "bit_identical_to_model_call": same})
        out["lanes"][str(lanes)] = rec
    out["one_batch_at_a_time_ms_per_step"] = round(1e3 * single, 4)
    best = max(out["lanes"].items(), key=lambda kv: kv[1]["Mtexels_per_s"])
    out["best"] = {"lanes": int(best[0]), **best[1]}
    return out


def cpu_baseline_worker(args):
    """Runs in a CHILD process (`bench.py --cpu-baseline-worker`): the CPU oracle forward on the host cores at the BENCHMARK'S
    OWN workload -- `--cpu-frames` frames of uv x uv UV (1024^2), k observation maps, cam = uv / 2, full forward + warp.
    One untimed pass at 64^2 pages the libraries in, then the first full-size pass is timed too: if it already takes longer
    than ~10 s it is the sample (a slower host must not take the bench over its budget), otherwise up to 5 passes / ~10 s more
    and the median."""
    import torch
    from oracle import nlt_oracle as O
    cores = os.cpu_count() or 1
    threads = min(cores, args.cpu_threads) if args.cpu_threads > 0 else cores
    torch.set_num_threads(threads)
    uv, n = args.uv, max(1, args.cpu_frames)
    cam = max(uv // 2, 32)
    with torch.no_grad():
        small = O.OracleModel(depth=args.depth, uvh=64, uvw=64, imh=32, imw=32, seed=0)
        sb, snn = O.synth_batch(1, 64, 64, 32, 32, 32, 32, k=args.k, seed=3)
        small.call(sb, 'test', nn_list=snn)
        om = O.OracleModel(depth=args.depth, uvh=uv, uvw=uv, imh=cam, imw=cam, seed=0)
        batch, nn = O.synth_batch(n, uv, uv, cam, cam, cam, cam, k=args.k, seed=3)
        t0 = time.perf_counter()
        om.call(batch, 'test', nn_list=nn)
        first = time.perf_counter() - t0
        times = []
        if first <= 10.0:
            t_start = time.perf_counter()
            while len(times) < 5 and (len(times) < 2 or time.perf_counter() - t_start < 10.0):
                t0 = time.perf_counter()
                om.call(batch, 'test', nn_list=nn)
                times.append(time.perf_counter() - t0)
    t = sorted(times)[len(times) // 2] if times else first
    how = ("median of %d passes after 1 full-size warm-up" % len(times)) if times else "the one (first) full-size pass"
    print(json.dumps({"value": round(n * uv * uv / t / 1e6, 3), "unit": "Mtexels/s", "cores": threads, "kind": "port",
                      "seconds_per_pass": round(t, 3), "frames": n, "uv": uv,
                      "sample": "CPU oracle (torch-CPU restatement of the TF2 path; TensorFlow is not installable "
                                "here), %d frame(s) %dx%d UV, k=%d, %dx%d camera, full forward + warp, %s, %d of %d host threads"
                                % (n, uv, uv, args.k, cam, cam, how, threads, cores)}), flush=True)


def cpu_baseline(args):
    """The CPU leg at the benchmark's own workload (BASELINE config 3: 1024^2 UV, k = 4), isolated in child processes with a hard
    timeout so that a slow or wedged host run can never take the GPU line with it.
      value: ALL host cores the way a CPU deployment of this data-parallel path would use them: one oracle
        process per 32 hardware threads, each rendering its own frame (N = 1) concurrently, texels/s summed.  (One process across
        256 threads collapses -- torch-CPU convs of this size stop scaling near 32 threads: 0.001 Mtexels/s measured.)
      one_process_4_frames: BASELINE.md's N = 4 figure, one process on 32 threads, run AFTER the concurrent leg (alone on the host)."""
    import subprocess
    cores = os.cpu_count() or 1
    per = 32 if args.cpu_threads == 0 else args.cpu_threads
    procs = max(1, cores // per) if args.cpu_threads == 0 else 1
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')

    def run(n_procs, frames, timeout):
        cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', '--uv', str(args.uv), '--k', str(args.k),
               '--depth', str(args.depth), '--cpu-threads', str(per), '--cpu-frames', str(frames)]
        ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, cwd=ROOT) for _ in range(n_procs)]
        recs, deadline = [], time.time() + timeout
        for p_ in ps:
            try:
                out, _ = p_.communicate(timeout=max(1.0, deadline - time.time()))
                lines = [l for l in out.decode().splitlines() if l.startswith('{')]
                recs.append(json.loads(lines[-1]))
            except Exception:                                      # timeout / crash of one worker: drop it, keep the rest
                p_.kill()
        return recs
    recs = run(procs, 1, 120)
    if not recs:
        return {"value": None, "unit": "Mtexels/s", "cores": 0, "kind": "port", "sample": "CPU oracle leg did not finish"}
    out = {"value": round(sum(r["value"] for r in recs), 3), "unit": "Mtexels/s", "cores": per * len(recs), "kind": "port",
           "processes": len(recs), "threads_per_process": per, "value_one_process": recs[0]["value"],
           "seconds_per_pass": recs[0].get("seconds_per_pass"),
           "sample": recs[0]["sample"] + "; %d such processes ran concurrently (one frame each), texels/s summed" % len(recs)}
    four = run(1, args.frames, 90)
    out["one_process_4_frames"] = ({k: four[0][k] for k in ("value", "unit", "cores", "seconds_per_pass", "sample")} if four else
                                   {"value": None, "sample": "did not finish within 90 s"})
    if four and four[0]["value"] > out["value"]:
        # the concurrent leg collapsed on this host (seen on some boxes: 35 s per pass instead of 5 with 8 x 32 threads): the
        # better CPU figure is the baseline -- one process, 32 threads, the bench's own 4-frame batch
        out["all_cores_concurrent"] = {k: out[k] for k in ("value", "cores", "processes", "threads_per_process", "seconds_per_pass")}
        out.update({"value": four[0]["value"], "cores": four[0]["cores"], "seconds_per_pass": four[0]["seconds_per_pass"],
                    "sample": four[0]["sample"] + "; (the %d-process all-cores leg gave LESS on this host: %.3f Mtexels/s summed)"
                              % (len(recs), out["all_cores_concurrent"]["value"])})
    return out


def bench_train(args, device, world, rank, n_steps, loss):
    """BASELINE config 4 per GPU: 4 frames, 1024^2 UV, k=1, Keras Adam-AMSGrad, the RCCL all-reduce(sum) of the flat fp32
    gradient bucket in three fixed ranges per step (two of them issued inside the backward); batches rotate through
    `Dataset.load_batch`'s staging ring.  Beside the headline at N = 1, one line per loss (the released configs train with
    `barron`); at N > 1 the first loss's line IS the top-level line (main)."""
    import torch
    import torch.distributed as dist
    from nlt_amd import trainvali
    from nlt_amd.models import get_model_class
    cfg, ds, id_lists = make_loader(args, device, 1, 'train', seed=200 + rank, loss=loss)
    model = get_model_class('nlt')(cfg).build(device)
    model.register_trainable()
    g = torch.Generator(device=device).manual_seed(4321)
    with torch.no_grad():
        for c in model._conv_layers():
            c.bias.uniform_(-0.1, 0.1, generator=g)
    opt = trainvali.make_optimizer(model, cfg)
    batches = [ds.load_batch(ids) for ids in id_lists]                 # one ring slot each: resident, distinct addresses
    gbs = world * args.frames
    tune_train = args.tune_cache + '.train' if args.tune_cache else None
    if tune_train and os.path.exists(tune_train):
        model.plan.load_tuning(tune_train)                           # the train plan's own tile choices (k = 1, keeps activations)
    step = trainvali.GraphedTrainStep(model, opt, gbs) if args.train_graph else trainvali.distributed_train_step
    run = (lambda b: step(b)) if args.train_graph else (lambda b: step(model, b, opt, gbs))
    for i in range(4 * len(batches)):                                # eager warm-ups (autotune, tapes of every batch), graph capture
        run(batches[i % len(batches)])
    if tune_train and not os.path.exists(tune_train) and rank == 0:
        model.plan.save_tuning(tune_train)
    if args.per_op_train and rank == 0:
        from nlt_amd.engine import OpTimer
        timer = OpTimer()
        model.plan.timer = timer
        for i in range(3):
            trainvali.distributed_train_step(model, batches[i % len(batches)], opt, gbs)
        rec = timer.collect()
        model.plan.timer = None
        table = sorted(((r[1] / r[0], l, timer.moved.get(l, r[2])) for l, r in rec.items()), reverse=True)
        tot = sum(t for t, _, _ in table)
        sys.stderr.write("train step (%s), plan launches only (loss / warp backward / Adam / all-reduce are outside the plan); "
                         "GB/s = bytes the launch itself moves\n" % loss)
        for t, l, nb in table[:60]:
            sys.stderr.write("%-22s %10.4f %6.1f%% %10.1f GB/s\n" % (l, t, 100 * t / tot, nb / t / 1e6))
        sys.stderr.write("sum of plan launches %.3f ms (%d launches)\n" % (tot, len(table)))
    replays0 = model.plan.tape_replays

    def timed(same_batch):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            loss_v, _ = run(batches[0 if same_batch else i % len(batches)])   # (`run` is looked up at call time)
        enq = time.perf_counter() - t0                               # host enqueue time (the GPU may still be running)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            te = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
        return el, enq, float(loss_v)
    el, enq, last = timed(False)
    replays = model.plan.tape_replays - replays0
    el_same, _, _ = timed(True)
    # what a multi-GPU run has to document about itself: the collective library's own world size, the two all-reduce
    # ranges, and the step with the first range issued from inside the backward plan (overlap) vs after it (serial)
    rg = list(model.bucket_ranges)
    comm = {"rccl_world": dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1,
            "allreduce_ranges": ["expanding blocks (issued inside the backward)", "encoder levels D..3 (issued inside the backward)",
                                 "levels 2..0 + head (after the backward)"],
            "allreduce_floats": [int(rg[i + 1] - rg[i]) for i in range(len(rg) - 1)],
            "allreduce_MB": [round(4e-6 * (rg[i + 1] - rg[i]), 2) for i in range(len(rg) - 1)],
            "backend": ("nccl (RCCL over xGMI)" if os.environ.get('NLT_BENCH_BACKEND', 'nccl') == 'nccl' else os.environ['NLT_BENCH_BACKEND'])
                       if world > 1 else "none (single rank: no collective is issued)",
            "scaling_curve": "never measured on hardware by the builder: no multi-GPU node was available to rounds 1-6 (SCALE_r01..r05 skipped); this run's own figures are comm.speedup_vs_one_rank / comm.scaling_efficiency"}
    if world > 1 and not args.train_graph:
        run_serial = lambda b: trainvali.distributed_train_step(model, b, opt, gbs, overlap=False)
        for i in range(len(batches)):
            run_serial(batches[i])
        run_saved, run = run, run_serial
        el_serial, _, _ = timed(False)
        run = run_saved
        comm["ms_per_step_overlapped"] = round(1e3 * el / n_steps, 3)
        comm["ms_per_step_serial"] = round(1e3 * el_serial / n_steps, 3)
        # the same step on every rank at once WITHOUT the collectives (forward + loss + backward + Adam on the local gradient):
        # what the all-reduces cost beyond what the backward hides.  Last: it lets the ranks' weights drift apart.
        def run_local(b):
            out_ = model.train_forward_backward(b, gbs)
            opt.step(model.flat_grads)
            return out_
        for i in range(len(batches)):
            run_local(batches[i])
        run_saved, run = run, run_local
        el_local, _, _ = timed(False)
        run = run_saved
        comm["ms_per_step_no_collective"] = round(1e3 * el_local / n_steps, 3)
        comm["exposed_ms"] = round(1e3 * (el - el_local) / n_steps, 3)
        # the scaling figure of THIS run: N ranks finish N x the frames of one rank in (overlapped) instead of (no collective)
        comm["speedup_vs_one_rank"] = round(world * el_local / el, 3)
        comm["scaling_efficiency"] = round(el_local / el, 4)
        if loss == args.train_loss.split(',')[0]:                   # (once per run: the leg does not depend on the loss)
            comm["light_events_ab"] = light_events_ab(args, device, world, rank, cfg, batches[0])
        comm["events"] = ("fenced (torch.cuda.Event) -- the default at world > 1 until `light_events_ab` has passed on real xGMI"
                          if not capi_light() else "light (no system-scope fence; NLT_LIGHT_EVENTS=1)")
    # per-rank host side of the step (what 8 single-threaded Python ranks on one host have to sustain): this rank's enqueue
    # time per step and the cores it may run on; gathered on rank 0
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = os.cpu_count() or 0
    per_rank = [{"rank": rank, "host_enqueue_ms_per_step": round(1e3 * enq / n_steps, 3), "cpus_allowed": affinity}]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
    comm["per_rank_host"] = per_rank
    flops = 3 * FLOP_PER_TEXEL[args.depth](1) * args.frames * args.uv * args.uv      # forward + backward-data + weight gradients
    return {"loss": loss, "comm": comm,
            "layerwise_flops_per_step_per_gpu": int(flops),
            "frac_of_fp32_mfma_peak": round(flops / (el / n_steps) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), "value": round(world * args.frames * args.uv * args.uv * n_steps / el / 1e6, 2), "unit": "Mtexels/s",
            "ms_per_step": round(1e3 * el / n_steps, 3), "ms_per_step_same_batch_every_step": round(1e3 * el_same / n_steps, 3),
            "host_enqueue_ms_per_step": round(1e3 * enq / n_steps, 3),
            "steps": n_steps, "global_batch": gbs, "distinct_batches_rotated": len(batches),
            "launch_tape_replays": int(replays), "plan_passes": 2 * n_steps,
            "launch": ("eager" if not args.train_graph or step.graph is None else
                       "hipGraph replay of forward + loss + backward; all-reduce and Adam-AMSGrad eager"),
            "graph_error": step.failed if args.train_graph else None,
            "workload": "BASELINE config 4: %d frames/GPU, %d^2 UV, k=1, loss %s, Adam-AMSGrad, flat %d-float "
                        "gradient bucket all-reduce; batches from Dataset.load_batch (uint8 store, staging ring)"
                        % (args.frames, args.uv, loss, model.flat_params.numel()),
            "final_loss": last}


def capi_light():
    from nlt_amd import capi
    return capi.light_events_enabled()


def light_events_ab(args, device, world, rank, cfg, batch, steps=6):
    """A/B of the two event kinds that order the backward plan's side streams -- among them the hand-over of a finished
    gradient range to the RCCL all-reduce -- at world > 1 (advisor r04, review r05 8c/8d).  Both legs: same weights, same batch,
    the SAME gradient w.r.t. the rendered texels (so the backward plan is bit-reproducible: its reductions are fixed-order),
    forward(train) + backward with ranges 0-1 all-reduced from inside the plan + range 2 after it, `steps` times (eager, recorded,
    replayed).  The all-reduced flat gradient buckets of the two legs must be BIT-IDENTICAL at every step; a missing fence
    would show as a stale range.  Times are the whole leg's, per step."""
    import torch
    import torch.distributed as dist
    from nlt_amd import capi
    from nlt_amd.models import get_model_class
    base, cvis, lvis, warp, nn_base, nn_rgb = batch[1], batch[2], batch[3], batch[4], batch[8], batch[9]
    if nn_rgb.dim() == 4:
        nn_rgb, nn_base = nn_rgb.unsqueeze(1), nn_base.unsqueeze(1)
    nn_rgb, nn_base = nn_rgb.contiguous(), nn_base.contiguous()
    g = torch.Generator(device=device).manual_seed(77)
    dpred = (torch.rand(tuple(base.shape), device=device, generator=g) - 0.5).contiguous()
    results, times, tuning = {}, {}, None
    model_ranges, weights = [], []
    saved = capi.LIGHT_EVENTS

    def one_leg(light, n_steps):
        capi.LIGHT_EVENTS = light                            # (events are created with the plan's side streams: a model per leg)
        model = get_model_class('nlt')(cfg).build(device)
        model.register_trainable()
        gw = torch.Generator(device=device).manual_seed(4321)
        with torch.no_grad():
            for c in model._conv_layers():
                c.bias.uniform_(-0.1, 0.1, generator=gw)
            if weights:                                      # the SAME weights in every leg (kernels are drawn per model instance)
                model.flat_params.copy_(weights[0])
            else:
                weights.append(model.flat_params.detach().clone())
        model.mark_weights_updated()
        if tuning is not None:
            model.plan.import_tuning(tuning)                 # the SAME plan-time choices in both legs: same kernels, same summation orders
        grad, r = model.flat_grads, model.bucket_ranges
        outs = []
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            works, sent = [], set()

            def reduce_range(i):
                if i not in sent and r[i + 1] > r[i]:
                    works.append(dist.all_reduce(grad[r[i]:r[i + 1]], op=dist.ReduceOp.SUM, async_op=True))
                sent.add(i)
            with torch.no_grad():
                model._render(base, cvis, lvis, warp, nn_rgb, nn_base, None, None, False, inference=False)
                grad.zero_()
                model.plan.grad_hook = reduce_range
                try:
                    model.plan.backward(dpred, base, cvis, lvis, nn_rgb, nn_base, None, generation=model.plan.generation)
                finally:
                    model.plan.grad_hook = None
            for i in range(len(r) - 1):
                reduce_range(i)
            for w_ in works:
                w_.wait()
            outs.append(grad.clone())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_steps
        tune = model.plan.export_tuning()
        model_ranges[:] = list(model.bucket_ranges)
        del model
        torch.cuda.empty_cache()
        return outs, dt, tune
    try:
        # plan-time trials once, rank 0's choices for every rank and both legs (two separately tuned plans sum in different orders)
        _, _, tune0 = one_leg(False, 1)
        box = [tune0]
        dist.broadcast_object_list(box, src=0)
        tuning = box[0]
        for leg, light in (("light", True), ("fenced", False)):
            outs, dt, _ = one_leg(light, steps)
            results[leg], times[leg] = outs, round(1e3 * dt, 3)
    finally:
        capi.LIGHT_EVENTS = saved
    same = all(torch.equal(a, b) for a, b in zip(results["light"], results["fenced"]))
    stable = all(torch.equal(a, results["fenced"][0]) for a in results["fenced"][1:])
    lstable = all(torch.equal(a, results["light"][0]) for a in results["light"][1:])
    flag = torch.tensor([int(same), int(stable), int(lstable)], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    rg = [int(x) for x in model_ranges]
    diff = [[float((a[rg[i]:rg[i + 1]] - b[rg[i]:rg[i + 1]]).abs().max()) if rg[i + 1] > rg[i] else 0.0 for i in range(len(rg) - 1)]
            for a, b in zip(results["light"], results["fenced"])]
    return {"steps_per_leg": steps, "bit_identical_light_vs_fenced_on_every_rank": bool(flag[0].item()),
            "fenced_leg_bit_reproducible_step_to_step": bool(flag[1].item()), "light_leg_bit_reproducible_step_to_step": bool(flag[2].item()),
            "max_abs_difference_per_step_and_range_on_rank_0": diff, "ms_per_step": times,
            "gradient_abs_max": float(results["fenced"][0].abs().max()),
            "what": "forward(train) + backward on a FIXED texel gradient with the 3-range RCCL all-reduce; flat buckets compared bitwise"}


def bench_config5(args, device):
    """BASELINE config 5 beside the headline (never instead of it): 2048^2 UV, 2 frames, k = 1, the SAME forward with the
    middle of the network fp32 (MFMA f32) and bf16 (v_mfma_f32_16x16x32_bf16, bf16-stored maps, fp32 accumulate)."""
    import copy
    import torch
    from nlt_amd.models import get_model_class
    a5 = copy.copy(args)
    a5.uv, a5.frames, a5.k, a5.batches, a5.store_frames = 2048, 2, 1, 3, 6
    out = {"workload": "BASELINE config 5: depth0 16/depth %d, 2 frames, 2048^2 UV, k=1, %d^2 camera warp" % (args.depth, args.cam)}
    preds = {}
    for prec in ('fp32', 'bf16'):
        cfg, ds, id_lists = make_loader(a5, device, 1, 'train', seed=500)
        cfg.set('DEFAULT', 'precision', prec)
        model = get_model_class('nlt')(cfg).build(device)
        g = torch.Generator(device=device).manual_seed(1234)
        for v in model.register_trainable() or model.trainable_variables:
            if v.dim() == 1:
                v.data.uniform_(-0.1, 0.1, generator=g)
        if preds:                                                # the SAME weights in both precisions (kernels are drawn per instance)
            with torch.no_grad():
                model.flat_params.copy_(weights)
            model.mark_weights_updated()
        else:
            weights = model.flat_params.detach().clone()
        batches = [ds.load_batch(ids) for ids in id_lists]
        for i in range(9):
            model.call(batches[i % 3], 'test')
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for i in range(n):
            model.call(batches[i % 3], 'test')
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        out[prec] = {"ms_per_step": round(1e3 * dt, 4), "Mtexels_per_s": round(a5.frames * a5.uv * a5.uv / dt / 1e6, 1),
                     "dtype": "f32" if prec == 'fp32' else "bf16 storage + bf16 MFMA (fp32 accumulate) for levels >= 3 and the "
                              "expanding blocks mirroring them; fp32 ends"}
        hbm, src = pmc_cfg5_bytes(prec)                          # whole-pass HBM bytes of THIS configuration (committed PMC passes)
        if hbm:
            out[prec].update({"hbm_bytes_per_step_pmc": hbm, "pmc_source": src,
                              "frac_of_hbm_peak": round(hbm / dt / 1e9 / HBM_PEAK_GBS, 4)})
        preds[prec] = model.call(batches[0], 'test')[3]['pred'].double()
        del model, batches, ds
        torch.cuda.empty_cache()
    # same weights (seeded), same batch: how far the bf16 middle moves the rendered texels, measured in THIS run on the GPU;
    # the comparison against the CPU oracle at this size is tests/test_gpu_baseline_sizes.py::test_config5_2048_fp32_and_bf16
    out["bf16"]["rel_l2_pred_vs_fp32_plan"] = float((preds['bf16'] - preds['fp32']).norm() / preds['fp32'].norm())
    out["bf16"]["oracle_parity"] = parity_record(('config5_2048_k1_n2_bf16', 'config5_2048_k1_n2_fp32'))
    return out


PRECISION_DTYPE = {
    'fp32': "f32 (every product on v_mfma_f32_16x16x4_f32 / fp32 VALU)",
    'f32x3_9': "f32 via 3xbf16 split, 9 exact products, fp32 accumulate (the LDS-tiled encoder convs: fp32 operands as three bf16 terms "
               "on v_mfma_f32_16x16x32_bf16, every term product exact in fp32); f32 storage everywhere, every other launch native f32",
    'f32x3': "f32 via 3xbf16 split, 6 of the 9 term products (the three of relative order 2^-24 dropped), fp32 accumulate; f32 storage",
    'bf16': "bf16 (fp32 accumulate) for the middle of the network, f32 ends",
}


def bench_f32_split(args, device, native, batches, precisions=('f32x3', 'f32x3_9')):
    """The SAME workload, weights and batches at the other precisions of the plan, beside the headline (VERDICT r03 ruling: the
    9-product split may be the headline when the native v_mfma_f32 line stays in the JSON as `native_fp32`; the 6-product form stays
    a sub-line).  rel-L2 of the rendered texels against the headline's plan is measured here; against the CPU oracle in
    tests/test_gpu_tile.py and tests/test_gpu_baseline_sizes.py."""
    import torch
    import nlt_amd
    from nlt_amd.models import get_model_class
    ref = native.call(batches[0], 'test')[3]['pred'].double()
    out = {"workload": "the headline's (BASELINE config 3), same weights and batches"}
    for prec in precisions:
        cfg = nlt_amd.make_config(depth=args.depth, uvh=args.uv, uvw=args.uv, imh=args.cam, imw=args.cam, bs=args.frames, precision=prec)
        model = get_model_class('nlt')(cfg).build(device)
        model.register_trainable()
        with torch.no_grad():
            model.flat_params.copy_(native.flat_params)
        model.mark_weights_updated()
        dt = time_forward(model, batches, 50)
        pred = model.call(batches[0], 'test')[3]['pred'].double()
        tiled = sorted(l for l, v in model.plan.lds_hints.items())
        out[prec] = {"ms_per_step": round(1e3 * dt, 4), "Mtexels_per_s": round(args.frames * args.uv * args.uv / dt / 1e6, 1),
                     "rel_l2_pred_vs_the_headline_plan": float((pred - ref).norm() / ref.norm()),
                     "dtype": PRECISION_DTYPE[prec],
                     "launches_on_the_lds_tiled_kernel": tiled, "launches_on_the_winograd_kernel": sorted(model.plan.wino_hints)}
        del model
        torch.cuda.empty_cache()
    return out


def bench_f32_split_pipelined(args, device, native, batches, lanes=4):
    """`precision = f32x3` AND several batches in flight: the two sub-line levers together (same weights and batches)."""
    import torch
    import nlt_amd
    from nlt_amd.models import get_model_class
    cfg = nlt_amd.make_config(depth=args.depth, uvh=args.uv, uvw=args.uv, imh=args.cam, imw=args.cam, bs=args.frames, precision='f32x3')
    model = get_model_class('nlt')(cfg).build(device)
    model.register_trainable()
    with torch.no_grad():
        model.flat_params.copy_(native.flat_params)
    model.mark_weights_updated()
    d1 = time_forward(model, batches, 40)
    dp, first = time_pipelined(model, batches, max(40, args.steps // 2), lanes, single=d1)
    rec = {"lanes": lanes, "one_batch_at_a_time_ms_per_step": round(1e3 * d1, 4), "ms_per_step": round(1e3 * dp, 4),
           "Mtexels_per_s": round(args.frames * args.uv * args.uv / dp / 1e6, 1)}
    if first is not None:
        rec["first_try_ms_per_step"] = round(1e3 * first, 4)
    del model
    torch.cuda.empty_cache()
    return rec


def parity_record(keys):
    """The newest committed HIP-vs-oracle figures for these test ids (profiles/*_parity_sizes.json, written by the -m gpu
    tests under NLT_PARITY_DUMP); bench.py itself never runs the oracle outside its cpu_baseline leg."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_parity_sizes.json')), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        got = {k_: d[k_] for k_ in keys if k_ in d}
        if got:
            got["source"] = "profiles/" + os.path.basename(path)
            return got
    return None


def identity_batches(n, uv, cam, k, device, nb=3):
    """`nb` float32 11-tuples of seeded uniform buffers; relight-only shapes (cam == uv) get the identity warp of
    nlt/README.md "Relighting Only?"."""
    import torch
    batches = [synth_device_batch(n, uv, cam, k, device, seed=900 + i) for i in range(nb)]
    if cam == uv:
        jj, ii = torch.meshgrid(torch.arange(cam, device=device), torch.arange(cam, device=device), indexing='xy')
        wp = torch.stack((jj / cam, ii / cam), -1)[None].repeat(n, 1, 1, 1).float().contiguous()
        batches = [b[:4] + (wp,) + b[5:] for b in batches]
    return batches


def time_forward(model, batches, steps):
    import torch
    for i in range(3 * len(batches)):                             # plan-time autotune, tape record, replays
        model.call(batches[i % len(batches)], 'test')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        model.call(batches[i % len(batches)], 'test')
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def time_pipelined(model, batches, steps, lanes, single=None, **mode):
    """(elapsed / steps, first try or None) with `lanes` batches in flight (nlt_amd.pipeline.RenderPipeline).  `single` = the
    one-batch-at-a-time time: a leg that comes out more than 15 % SLOWER than that is the fresh-box state DESIGN.md section 4b
    describes; it is measured once more after 10 s of sustained single-lane steps and both figures are returned."""
    import torch
    from nlt_amd.pipeline import RenderPipeline
    pipe = RenderPipeline(model, lanes, **mode)
    for _ in range(3):
        for t in [pipe.submit(batches[i % len(batches)], 'test') for i in range(2 * lanes)]:
            t.result()

    def timed():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tickets = [pipe.submit(batches[i % len(batches)], 'test') for i in range(steps)]
        for t in tickets[-lanes:]:
            t.result()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps
    dt, first = timed(), None
    if single is not None and dt > 1.15 * single:
        first = dt
        t_end = time.perf_counter() + 10.0
        while time.perf_counter() < t_end:
            for i in range(50):
                model.call(batches[i % len(batches)], 'test')
            torch.cuda.synchronize()
        dt = timed()
    pipe.close()
    return dt, first


def bench_released_shapes(args, device, world, rank):
    """The two shapes the reference releases configs for, beside the headline: BASELINE config 2 (dragon_specular.ini: depth
    256, 512^2 UV, bs 4, k = 1, relight only: identity warp) forward AND train step with the released loss (barron), and
    BASELINE config 1 (dragon_sss.ini: depth 1024, 256^2) forward; each with its fraction of the fp32 MFMA roof by SURVEY
    8d's layer-wise FLOP, and how the forward's throughput grows with frames per call (nlt_test.py --batch_size_override,
    nlt/nlt_test.py:33-42,78-94): these shapes are latency-bound at 4 frames."""
    import copy
    import torch
    import nlt_amd
    from nlt_amd.models import get_model_class
    out = {}
    for name, depth, uv in (("config2_512_depth256", 256, 512), ("config1_256_depth1024", 1024, 256)):
        cfg = nlt_amd.make_config(depth=depth, uvh=uv, uvw=uv, imh=uv, imw=uv, bs=4)
        model = get_model_class('nlt')(cfg).build(device)
        g = torch.Generator(device=device).manual_seed(1234)
        for v in model.register_trainable() or model.trainable_variables:
            if v.dim() == 1:
                v.data.uniform_(-0.1, 0.1, generator=g)
        rec = {"workload": "depth0 16/depth %d, %d^2 UV, k=1, identity warp (relight only), fp32" % (depth, uv), "forward": {}}
        for frames in (4, 16, 64):
            batches = identity_batches(frames, uv, uv, 1, device)
            dt = time_forward(model, batches, 60 if frames <= 16 else 20)
            fl = FLOP_PER_TEXEL[depth](1) * frames * uv * uv
            rec["forward"]["%d_frames" % frames] = {
                "ms_per_step": round(1e3 * dt, 4), "Mtexels_per_s": round(frames * uv * uv / dt / 1e6, 1),
                "frac_of_fp32_mfma_peak": round(fl / dt / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}
            del batches
            torch.cuda.empty_cache()
        out[name] = rec
        del model
        torch.cuda.empty_cache()
    a2 = copy.copy(args)
    a2.uv, a2.cam, a2.frames, a2.k, a2.depth, a2.per_op_train, a2.tune_cache = 512, 512, 4, 1, 256, False, None
    t = bench_train(a2, device, world, rank, 50, 'barron')
    out["config2_512_depth256"]["train_step_barron_bs4"] = {k_: t[k_] for k_ in (
        "ms_per_step", "value", "unit", "frac_of_fp32_mfma_peak", "layerwise_flops_per_step_per_gpu", "host_enqueue_ms_per_step",
        "launch_tape_replays", "final_loss")}
    return out


def bench_infer_mode(args, device):
    """Sub-line `nlt_test_infer`: the reference's own way of rendering a test set (nlt/nlt_test.py:78-127) -- `extract_feat`
    averages every level's observation features over the training frames once, `infer` then calls
    `Model.call(batch, 'test', obs_override=feat_agg)` per batch -- on the fused query-only plan (engine_infer.py,
    csrc/front_ovr.hip), at BASELINE config 3's UV size and config 2's, 4 frames per batch like the released configs,
    feat_agg from 8 training frames.  One step = one Model.call (network + UV -> camera warp) over one of three rotating
    batches resident in HBM; fp32 (every product on v_mfma_f32).  Beside it the general layer-by-layer plan on the same
    inputs (what rounds 1-5 ran for this mode) and two batches in flight."""
    import torch
    import nlt_amd
    from nlt_amd import nlt_test
    from nlt_amd.engine import OpTimer
    from nlt_amd.models import get_model_class
    out = {"what": "nlt_test.extract_feat + nlt_test.infer: Model.call(batch, 'test', obs_override=feat_agg), 4 frames per batch, "
                   "feat_agg from 8 training frames, fp32"}
    for name, uv, cam in (("uv1024_cam512", 1024, 512), ("uv512_identity_warp", 512, 512)):
        cfg = nlt_amd.make_config(depth=args.depth, uvh=uv, uvw=uv, imh=cam, imw=cam, bs=4)
        model = get_model_class('nlt')(cfg).build(device)
        g = torch.Generator(device=device).manual_seed(1234)
        for v in model.register_trainable() or model.trainable_variables:
            if v.dim() == 1:
                v.data.uniform_(-0.1, 0.1, generator=g)
        def loader_batches(frames, seed, nb=3):
            # the headline's data source: Dataset.load_batch on a seeded synthetic uint8 capture store, chart-structured uv2cam map
            # (relight-only shapes, cam == uv: the identity warp of nlt/README.md "Relighting Only?")
            if cam == uv:
                return identity_batches(frames, uv, cam, 1, device, nb=nb)
            import copy
            a2 = copy.copy(args)
            a2.uv, a2.cam, a2.frames, a2.k, a2.batches, a2.store_frames = uv, cam, frames, 1, nb, nb * frames
            _, ds_, ids_ = make_loader(a2, device, 1, 'train', seed=seed)
            loaders[frames] = (ds_, ids_[:nb])
            return [ds_.load_batch(i_) for i_ in ids_[:nb]]
        loaders = {}
        train = loader_batches(4, 700)[:2]
        batches = loader_batches(4, 900)
        for _ in range(3):                                            # (allocator + first-launch warm-up)
            agg = nlt_test.extract_feat(model, train)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            agg = nlt_test.extract_feat(model, train)
        torch.cuda.synchronize()
        t_feat = (time.perf_counter() - t0) / 3
        del train

        def run(steps, **kw):
            for i in range(3 * len(batches)):
                model.call(batches[i % len(batches)], 'test', obs_override=agg)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(steps):
                model.call(batches[i % len(batches)], 'test', obs_override=agg)
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / steps
        t_first0 = time.perf_counter()
        model.call(batches[0], 'test', obs_override=agg)
        torch.cuda.synchronize()
        t_first = time.perf_counter() - t_first0
        dt = run(max(20, args.steps))
        replays = int(model.plan.tape_replays)
        with_loader = None
        if 4 in loaders:                                              # the render loop as nlt_test.infer runs it: the loader inside
            ds_, ids_ = loaders[4]
            for i in range(2 * len(ids_)):
                model.call(ds_.load_batch(ids_[i % len(ids_)]), 'test', obs_override=agg)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nl = max(20, min(args.steps, 50))
            for i in range(nl):
                model.call(ds_.load_batch(ids_[i % len(ids_)]), 'test', obs_override=agg)
            torch.cuda.synchronize()
            dl = (time.perf_counter() - t1) / nl
            with_loader = {"ms_per_step": round(1e3 * dl, 4), "Mtexels_per_s": round(4 * uv * uv / dl / 1e6, 1),
                           "what": "Dataset.load_batch (uint8 store -> float32 11-tuple, staging ring) inside the timed loop"}
            try:                                                      # ... and with the texel buffers left in the uint8 store
                for i in range(2 * len(ids_)):
                    model.call(ds_.load_batch(ids_[i % len(ids_)], resident=True), 'test', obs_override=agg)
                torch.cuda.synchronize()
                r_before = int(model.plan.tape_replays)
                t1 = time.perf_counter()
                for i in range(nl):
                    model.call(ds_.load_batch(ids_[i % len(ids_)], resident=True), 'test', obs_override=agg)
                torch.cuda.synchronize()
                dr = (time.perf_counter() - t1) / nl
                with_loader_resident = {
                    "ms_per_step": round(1e3 * dr, 4), "Mtexels_per_s": round(4 * uv * uv / dr / 1e6, 1),
                    "launch_tape_replays_in_the_timed_loop": int(model.plan.tape_replays) - r_before,
                    "what": "Dataset.load_batch(resident=True) inside the timed loop: frame ids only; nlt_front_ovr_forward_u8 and "
                            "nlt_warp_forward_store read the uint8 / fp16 capture store in place (no float batch is assembled)"}
            except Exception as e:
                with_loader_resident = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        timer = OpTimer()
        model.plan.timer = timer
        for i in range(3):
            model.call(batches[i % len(batches)], 'test', obs_override=agg)
        rec_ = timer.collect()
        model.plan.timer = None
        flops = sum(timer.flops.get(l, 0) for l in rec_)
        moved = sum(timer.moved.get(l, r[2]) for l, r in rec_.items())
        table = sorted(((r[1] / r[0], l) for l, r in rec_.items()), reverse=True)
        dom_ms, dom = table[0]
        texels = 4 * uv * uv
        rec = {"ms_per_step": round(1e3 * dt, 4), "Mtexels_per_s": round(texels / dt / 1e6, 1), "launch_tape_replays": replays,
               "plan_launches": len(rec_), "useful_flops_per_step": int(flops), "flop_per_texel": round(flops / texels, 1),
               "frac_of_fp32_mfma_peak": round(flops / dt / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
               "plan_algorithmic_bytes_per_step": int(moved), "frac_of_hbm_peak_algorithmic": round(moved / dt / 1e9 / HBM_PEAK_GBS, 4),
               "dominant_launch": {"label": dom, "ms": round(dom_ms, 4), "TFLOPs": round(timer.flops.get(dom, 0) / dom_ms / 1e9, 2),
                                   "GBps_of_its_own_traffic": round(timer.moved.get(dom, rec_[dom][2]) / dom_ms / 1e6, 1)},
               "extract_feat_8_frames_ms": round(1e3 * t_feat, 3),
               "extract_feat_Mtexels_per_s": round(8 * uv * uv / t_feat / 1e6, 1),
               "first_call_ms_override_maps_and_plan_time_trials": round(1e3 * t_first, 1)}
        if with_loader:
            rec["including_loader_float32_batch_assembled_per_step"] = with_loader
            rec["including_loader_uint8_store_resident"] = with_loader_resident
        if name == "uv1024_cam512":
            try:
                from nlt_amd.pipeline import RenderPipeline
                with RenderPipeline(model, 2) as pipe:
                    seq = [batches[i % len(batches)] for i in range(max(20, args.steps))]
                    pipe.render(seq[:12], 'test', obs_override=agg)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    pipe.render(seq, 'test', obs_override=agg)
                    torch.cuda.synchronize()
                    d2 = (time.perf_counter() - t1) / len(seq)
                rec["two_batches_in_flight"] = {"ms_per_step": round(1e3 * d2, 4), "Mtexels_per_s": round(texels / d2 / 1e6, 1)}
            except Exception as e:
                rec["two_batches_in_flight"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            # frames per call (nlt_test.py --batch_size_override): the mid-network launches of a 4-frame batch are latency-bound
            try:
                b16 = loader_batches(16, 950)
                for i in range(6):
                    model.call(b16[i % 3], 'test', obs_override=agg)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                n16 = 20
                for i in range(n16):
                    model.call(b16[i % 3], 'test', obs_override=agg)
                torch.cuda.synchronize()
                d16 = (time.perf_counter() - t1) / n16
                rec["16_frames_per_call"] = {"ms_per_step": round(1e3 * d16, 4), "Mtexels_per_s": round(16 * uv * uv / d16 / 1e6, 1)}
                del b16
                torch.cuda.empty_cache()
                for i in range(3):                                    # (back on the 4-frame buffers / tapes)
                    model.call(batches[i], 'test', obs_override=agg)
            except Exception as e:
                rec["16_frames_per_call"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            # the headline's precision (three-term split, 9 exact products, on the LDS-tiled stride-1 encoder convs) on the same weights
            try:
                ref_pred = model.call(batches[0], 'test', obs_override=agg)[3]['pred'].double()
                cfg9 = nlt_amd.make_config(depth=args.depth, uvh=uv, uvw=uv, imh=cam, imw=cam, bs=4, precision='f32x3_9')
                m9 = get_model_class('nlt')(cfg9).build(device)
                m9.register_trainable()
                with torch.no_grad():
                    m9.flat_params.copy_(model.flat_params)
                m9.mark_weights_updated()
                for i in range(3 * len(batches)):
                    m9.call(batches[i % len(batches)], 'test', obs_override=agg)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                n9 = max(20, args.steps)
                for i in range(n9):
                    m9.call(batches[i % len(batches)], 'test', obs_override=agg)
                torch.cuda.synchronize()
                d9 = (time.perf_counter() - t1) / n9
                p9 = m9.call(batches[0], 'test', obs_override=agg)[3]['pred'].double()
                rec["f32x3_9"] = {"ms_per_step": round(1e3 * d9, 4), "Mtexels_per_s": round(texels / d9 / 1e6, 1),
                                  "rel_l2_pred_vs_fp32": float((p9 - ref_pred).norm() / ref_pred.norm()),
                                  "launches_on_the_three_term_kernel": sorted(l for l in m9.plan.lds_hints if l.endswith('.q.s1'))}
                del m9
                torch.cuda.empty_cache()
            except Exception as e:
                rec["f32x3_9"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            model.plan.fuse_override = False
            dg = run(10)
            model.plan.fuse_override = True
            rec["general_layer_by_layer_plan"] = {"ms_per_step": round(1e3 * dg, 4), "Mtexels_per_s": round(texels / dg / 1e6, 1),
                                                  "note": "NLT_FUSED_OVERRIDE=0: the plan rounds 1-5 ran this mode on (also serves per-frame override maps)"}
        out[name] = rec
        del model, batches, agg
        torch.cuda.empty_cache()
    return out


def bench_released_pipelined(args, device):
    """The released shapes at the render loop's own batch size (4 frames) with several batches in flight
    (nlt_amd.pipeline.RenderPipeline).  At these shapes one batch is a 28-deep chain of 10-30 us launches that leaves most of
    the chip idle AND costs the host as long to enqueue as the GPU needs to run it, so the lanes replay one hipGraph each
    (single-stream lanes); `eager_4_lanes` = launch tapes for comparison.  Runs in its own process (released_pipelined_child)."""
    import torch
    import nlt_amd
    from nlt_amd.models import get_model_class
    out = {"launch": "hipGraph replay per lane (single-stream lanes); eager_4_lanes: launch tapes, two streams per lane"}
    for name, depth, uv in (("config2_512_depth256", 256, 512), ("config1_256_depth1024", 1024, 256)):
        cfg = nlt_amd.make_config(depth=depth, uvh=uv, uvw=uv, imh=uv, imw=uv, bs=4)
        model = get_model_class('nlt')(cfg).build(device)
        g = torch.Generator(device=device).manual_seed(1234)
        for v in model.register_trainable() or model.trainable_variables:
            if v.dim() == 1:
                v.data.uniform_(-0.1, 0.1, generator=g)
        batches = identity_batches(4, uv, uv, 1, device)
        fl = FLOP_PER_TEXEL[depth](1) * 4 * uv * uv
        d1 = time_forward(model, batches, 120)
        rec = {"one_batch_at_a_time": {"ms_per_step": round(1e3 * d1, 4), "Mtexels_per_s": round(4 * uv * uv / d1 / 1e6, 1)}}
        for lanes, mode in ((2, {'graphs': True}), (4, {'graphs': True}), (8, {'graphs': True}), (4, {})):
            dp, first = time_pipelined(model, batches, 120, lanes, single=d1, **mode)
            r = {"ms_per_step": round(1e3 * dp, 4), "Mtexels_per_s": round(4 * uv * uv / dp / 1e6, 1),
                 "frac_of_fp32_mfma_peak": round(fl / dp / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}
            if first is not None:
                r["first_try_ms_per_step"] = round(1e3 * first, 4)
                r["re_measured_after"] = "10 s of sustained single-lane steps"
            rec["%s%d_lanes" % ('' if mode else 'eager_', lanes)] = r
        out[name] = rec
        del model, batches
        torch.cuda.empty_cache()
    return out


def child_leg(flag, timeout=420, extra=()):
    """One sub-line measured in a FRESH process on the same GPU, with a hard timeout: the small, latency-bound shapes are
    sensitive to what the bench process has accumulated by the time it reaches them (dozens of HIP streams, graph pools,
    allocator fragmentation): config 1's forward measures 0.40 ms in a clean process and 0.52 ms late in this one."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), flag] + list(extra)
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=ROOT, timeout=timeout).stdout.decode()
        d = json.loads([l for l in out.splitlines() if l.startswith('{')][-1])
        d["measured_in"] = "a child process of the bench on the same GPU (clean HIP state)"
        return d
    except Exception as e:                                         # a sub-line: never take the headline with it
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def released_pipelined_child(args):
    """`bench_released_pipelined` in a fresh process on the same GPU (this process' earlier legs leave dozens of HIP streams and
    graph pools behind, and the small shapes are sensitive to that: one batch at a time 0.44 ms in a clean process, 0.57 ms
    at the end of this one), with a hard timeout."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--released-pipelined-worker']
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=ROOT, timeout=240).stdout.decode()
        return json.loads([l for l in out.splitlines() if l.startswith('{')][-1])
    except Exception as e:                                         # a sub-line: never take the headline with it
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def bench_stress_64ch(device):
    """north_star's "1024^2 UV x 64-ch" point: no tensor of the released net is 64 channels wide at full
    resolution, so this is the channel-width stress of the per-texel kernels, reported beside the exact shapes: 1x1
    channel mix 64 -> 64 (fp32 MFMA and bf16 MFMA), the same 1x1 on a virtual concat (32 | 32 -> 64), and the resampler
    on a 64-channel map (1024^2 -> 512^2 camera pixels, chart-structured map).  One frame of 1024^2 texels; GB/s =
    compulsory bytes (each input element read once, each output element written once) / time, fraction of 8 TB/s."""
    import torch
    from nlt_amd import capi as C
    from nlt_amd.datasets.synth import chart_warp
    n, uv, c = 1, 1024, 64
    g = torch.Generator(device=device).manual_seed(7)
    x = torch.rand((n, uv, uv, c), device=device, generator=g)
    w = (torch.rand((1, 1, c, c), device=device, generator=g) - 0.5) * 0.3
    b = torch.rand(c, device=device, generator=g) - 0.5
    y = torch.empty_like(x)
    packed = C.pack_conv_weights(C.CONV1X1, w, c, 0, c)
    packed2 = C.pack_conv_weights(C.CONV1X1, w, c // 2, c // 2, c)
    xa, xb = x[..., :c // 2].contiguous(), x[..., c // 2:].contiguous()
    xh = x.to(torch.bfloat16)
    pb = C.chmix_bf16_pack(w)
    warp_px = (chart_warp(n, 512, g, device).float() * uv).contiguous()

    def timeit(fn, reps=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    texels = n * uv * uv
    # the fp32 1x1s run on the network's own implicit-GEMM kernel with the wave tile chosen by timing, as the plan does
    def conv1x1(hint, concat):
        if concat:
            return lambda: C.conv_forward(C.CONV1X1, xa, c // 2, c // 2, xb, c // 2, c // 2, n, uv, uv, w, packed2, b, c, y, c,
                                          act=True, alpha=0.3, algo=C.ALGO_MFMA, tile_hint=hint)
        return lambda: C.conv_forward(C.CONV1X1, x, c, c, None, 0, 0, n, uv, uv, w, packed, b, c, y, c, act=True, alpha=0.3,
                                      algo=C.ALGO_MFMA, tile_hint=hint)
    best = {}
    for concat in (False, True):
        tried = [(timeit(conv1x1(16 * r + ct, concat), reps=8), 16 * r + ct) for r in (1, 2, 4) for ct in (1, 2, 4)]
        best[concat] = min(tried)[1]
    cam_px = 512 * 512 * n
    legs = {
        "conv1x1_f32_64to64": (conv1x1(best[False], False), 4 * 2 * c * texels, 2 * c * c * texels),
        "conv1x1_f32_virtual_concat_32_32to64": (conv1x1(best[True], True), 4 * 2 * c * texels, 2 * c * c * texels),
        "conv1x1_bf16_64to64": (lambda: C.chmix_bf16_forward(xh, pb, b, c, act=True, alpha=0.3), 2 * 2 * c * texels, 2 * c * c * texels),
        # compulsory bytes of the gather: the map (8 B), ONE new texel per camera pixel (the other taps are shared with the
        # neighbouring pixels of a chart and cache-served), the output
        "resampler_f32_64ch_to_512cam": (lambda: C.resample_forward(x, warp_px), cam_px * (8 + 4 * c + 4 * c), 0),
    }
    out = {"workload": "1 frame, 1024^2 UV, 64 channels (NHWC)",
           "conv1x1_f32_wave_tile": {"plain": "%dx%d" % (best[False] >> 4, best[False] & 15), "virtual_concat": "%dx%d" % (best[True] >> 4, best[True] & 15)}}
    for name, (fn, nbytes, flops) in legs.items():
        dt = timeit(fn)
        out[name] = {"ms": round(1e3 * dt, 4), "Gtexels_per_s": round(texels / dt / 1e9, 2), "GBps": round(nbytes / dt / 1e9, 1),
                     "frac_of_hbm_peak": round(nbytes / dt / 1e9 / HBM_PEAK_GBS, 4)}
        if flops:
            out[name]["TFLOPs"] = round(flops / dt / 1e12, 1)
    r = out["resampler_f32_64ch_to_512cam"]
    r["tap_GBps_cache_served"] = round(cam_px * (8 + 4 * 4 * c + 4 * c) / (r["ms"] * 1e-3) / 1e9, 1)   # all four taps counted
    return out


def relaunch_multi_rank(n):
    """`python bench.py --gpus N` (N > 1) WITHOUT a torchrun environment: one rank per GPU is the only way this line means
    anything, so the process re-executes itself under `python -m torch.distributed.run` (the launcher the driver uses) and
    hands its exit status on -- it never runs one rank silently and prints `n_gpus: 1` (review r05, weak point 8a)."""
    import socket
    import subprocess
    import torch
    share = os.environ.get('NLT_BENCH_SHARE_GPU', '0') == '1'
    have = torch.cuda.device_count()
    if have < n and not share:
        sys.stderr.write("bench.py --gpus %d: this node shows %d GPU(s); refusing to run fewer ranks than asked for "
                         "(NLT_BENCH_SHARE_GPU=1 + NLT_BENCH_BACKEND=gloo: a dry run of the code path on one GPU)\n" % (n, have))
        sys.exit(2)
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: what RCCL needs on this pool
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without WORLD_SIZE: re-executing as `%s`\n" % (n, ' '.join(cmd[1:9]) + ' bench.py ...'))
    sys.exit(subprocess.call(cmd, env=env))


def check_ranks(args, dist, torch, world, rank, local_rank, device):
    """What a multi-rank run has to prove about itself before any number is believed: the process group spans exactly --gpus
    ranks, every rank drives its OWN device (unless this is the declared one-GPU dry run), and a collective of the backend
    really sums over all of them."""
    import socket
    if dist.get_world_size() != args.gpus:
        raise SystemExit("process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))
    props = torch.cuda.get_device_properties(device)
    me = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), "pid": os.getpid(),
          "device_index": torch.cuda.current_device(), "device_uuid": str(getattr(props, 'uuid', '')), "device_name": props.name}
    ranks = [None] * world
    dist.all_gather_object(ranks, me)
    share = os.environ.get('NLT_BENCH_SHARE_GPU', '0') == '1'
    distinct = len({(r["host"], r["device_uuid"] or r["device_index"]) for r in ranks})
    if distinct != world and not share:
        raise SystemExit("%d ranks drive %d distinct devices: one rank per GPU is required (LOCAL_RANK -> cuda:LOCAL_RANK)" % (world, distinct))
    ones = torch.ones(1, device=device)
    dist.all_reduce(ones)
    if int(ones.item()) != world:
        raise SystemExit("all_reduce of ones over the process group gives %d, not %d" % (int(ones.item()), world))
    try:
        ver = '.'.join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        ver = None
    return {"process_group_backend": dist.get_backend(), "process_group_world_size": dist.get_world_size(),
            "allreduce_of_ones": int(ones.item()), "rccl_version": ver, "distinct_devices": distinct, "ranks": ranks}


def main():
    args = parse()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args)
    if args.released_pipelined_worker:
        import torch
        print(json.dumps(bench_released_pipelined(args, torch.device('cuda', 0))), flush=True)
        return
    if args.released_shapes_worker:
        import torch
        torch.cuda.set_device(0)
        print(json.dumps(bench_released_shapes(args, torch.device('cuda', 0), 1, 0)), flush=True)
        return
    if args.infer_mode_worker:
        import torch
        torch.cuda.set_device(0)
        print(json.dumps(bench_infer_mode(args, torch.device('cuda', 0))), flush=True)
        return
    import torch
    import torch.distributed as dist
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        relaunch_multi_rank(args.gpus)                              # (does not return)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # NLT_BENCH_BACKEND=gloo + NLT_BENCH_SHARE_GPU=1: a DRY RUN of the multi-rank line on a one-GPU box (every rank on cuda:0,
    # collectives through the host) -- exercises the N > 1 code path, measures nothing; the line says so (`config.dry_run`).
    backend = os.environ.get('NLT_BENCH_BACKEND', 'nccl')
    if os.environ.get('NLT_BENCH_SHARE_GPU', '0') == '1':
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)
    rank_check = check_ranks(args, dist, torch, world, rank, local_rank, device) if world > 1 else None

    import nlt_amd
    from nlt_amd import capi
    from nlt_amd.engine import OpTimer
    from nlt_amd.models import get_model_class
    cfg, ds, id_lists = make_loader(args, device, args.k, 'train', seed=100 + rank)
    cfg.set('DEFAULT', 'precision', args.precision)
    model = get_model_class('nlt')(cfg).build(device)
    # random-init weights of the released architecture; non-zero biases so the bias path is live
    g = torch.Generator(device=device).manual_seed(1234)          # same weights on every rank
    for v in model.register_trainable() or model.trainable_variables:
        if v.dim() == 1:
            v.data.uniform_(-0.1, 0.1, generator=g)
    if args.algo == 'direct':
        model.conv_algo = capi.ALGO_DIRECT
    if args.no_fused:
        model.plan.fuse_ends = False
    model.use_graphs = bool(args.graph)
    if args.tune_cache and os.path.exists(args.tune_cache):
        model.plan.load_tuning(args.tune_cache)
    # `--batches` different batches, assembled by Dataset.load_batch into its staging ring (one slot each): inputs are
    # resident in HBM as the float32 11-tuple the reference's `_load_data` hands to Model.call
    batches = [ds.load_batch(ids) for ids in id_lists]
    calls = [0]

    def step():
        calls[0] += 1
        return model.call(batches[calls[0] % len(batches)], 'test')

    # per-launch survey (outside the timed region) -> dominant kernel
    for _ in range(2):
        step()
    if args.tune_cache and not os.path.exists(args.tune_cache) and rank == 0:
        model.plan.save_tuning(args.tune_cache)
    timer = OpTimer()
    model.plan.timer = timer
    for _ in range(3):
        step()
    rec = timer.collect()
    model.plan.timer = None
    table = sorted(((r[1] / r[0], l, r[2]) for l, r in rec.items()), reverse=True)
    if args.per_op and rank == 0:
        tot = sum(t for t, _, _ in table)
        sys.stderr.write("%-12s %10s %7s %10s %8s\n" % ("launch", "ms", "%", "GB/s", "TFLOP/s"))
        for t, l, nb in table:
            sys.stderr.write("%-12s %10.4f %6.1f%% %10.1f %8.1f\n" % (l, t, 100 * t / tot, timer.moved.get(l, nb) / t / 1e6,
                                                                     timer.flops.get(l, 0) / t / 1e9))
        sys.stderr.write("sum of launches %.3f ms\n" % tot)
    dominant = args.dominant or table[0][1]

    for _ in range(args.warmup):
        step()
    dom = OpTimer(); dom.only = {dominant}
    model.plan.timer = None if args.graph else dom      # events cannot be recorded inside a replayed graph
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if args.graph:                                      # same steps, eager, only to time the dominant launch
        model.use_graphs = False
        model.plan.timer = dom
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        model.use_graphs = True
    model.plan.timer = None
    if world > 1:
        te = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    drec = dom.collect()[dominant]
    dom_ms = drec[1] / drec[0]
    dom_layerwise = drec[2]                                   # SURVEY 8d layer-wise bytes of what the launch computes
    dom_bytes = timer.moved.get(dominant, dom_layerwise)      # a fused launch: the bytes it must itself move
    dom_flops = timer.flops.get(dominant, 0)

    # beside the headline: the same forward INCLUDING the loader, fed (a) the way the reference is (float32 batch assembled
    # per step) and (b) store-resident (uint8 frame ids; the front kernel converts in registers)
    with_loader = {}
    if not args.graph and not args.headline_only:
        n_l = max(10, min(args.steps, 50))
        for name, resident in (("float32_batch_assembled_per_step", False), ("uint8_store_resident", True)):
            for i in range(2 * len(id_lists)):
                model.call(ds.load_batch(id_lists[i % len(id_lists)], resident=resident), 'test')
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(n_l):
                model.call(ds.load_batch(id_lists[i % len(id_lists)], resident=resident), 'test')
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            with_loader[name] = {"ms_per_step": round(1e3 * dt / n_l, 4),
                                 "Mtexels_per_s_per_gpu": round(args.frames * args.uv * args.uv * n_l / dt / 1e6, 1)}
    fwd_replays = int(model.plan.tape_replays)

    train = []
    n_train = max(50, args.steps // 2) if args.train_steps < 0 else args.train_steps
    if world > 1:
        n_train = args.steps            # the data-parallel line IS the train step: exactly --steps timed steps
    if n_train > 0 and not args.headline_only:
        for loss in args.train_loss.split(','):
            train.append(bench_train(args, device, world, rank, n_train, loss))

    released = None
    if not args.headline_only and not args.no_released_shapes and args.uv == 1024:
        if world == 1:
            released = child_leg('--released-shapes-worker', extra=['--steps', str(args.steps), '--depth', str(args.depth)])
            if "error" in released:
                released = bench_released_shapes(args, device, world, rank)
        else:
            released = bench_released_shapes(args, device, world, rank)     # (its train leg is collective on every rank)

    if rank == 0:
        texels = world * args.frames * args.uv * args.uv * args.steps
        value = texels / elapsed / 1e6
        ach = dom_bytes / (dom_ms * 1e-3) / 1e9
        bpt = algorithmic_bytes_per_texel(args.k)
        # HBM bytes per launch of the dominant kernel: NOT measured in this run (PMC needs rocprofv3 around the process) --
        # read from the newest committed counter pass; `traffic_source` names the file so a stale figure is visible
        traffic, traffic_src = pmc_traffic(dominant)
        if traffic_src is not None:
            traffic_src = "profiles/%s (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not this run)" % traffic_src
        traffic_committed, live_note, live_step_bytes = traffic, None, None
        if world == 1 and not args.headline_only and not args.no_live_pmc and not args.graph:
            import tempfile
            with tempfile.NamedTemporaryFile(suffix='.json', dir='/tmp', delete=False) as tf_:
                tune_tmp = tf_.name
            try:
                os.unlink(tune_tmp)
                model.plan.save_tuning(tune_tmp)
                live, live_note = live_pmc_traffic(dominant, args, tune_tmp)
            finally:
                if os.path.exists(tune_tmp):
                    os.unlink(tune_tmp)
            if live is not None:
                live_step_bytes = live['step']
                traffic = live['dominant']
                traffic_src = ("measured in this run: two child passes of the timed forward under rocprofv3 --pmc FETCH_SIZE / "
                               "--pmc WRITE_SIZE (FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE as reported)")
        if dom_flops / max(dom_bytes, 1) > MFMA_F32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):   # above the fp32 ridge
            tf = dom_flops / (dom_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": dominant, "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "launch_ms": round(dom_ms, 4), "flops_per_launch": int(dom_flops),
                    "algorithmic_bytes_per_launch": int(dom_bytes)}
        else:
            roof = {"bound": "hbm", "kernel": dominant, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "launch_ms": round(dom_ms, 4), "algorithmic_bytes_per_launch": int(dom_bytes)}
        if traffic is not traffic_committed or live_note:
            roof["traffic_committed_profile"] = traffic_committed
            if live_note:
                roof["traffic_live_pass"] = "not taken: " + live_note
        if dom_layerwise != dom_bytes:                          # fused launch: also what it replaces, layer by layer
            roof["layerwise_bytes_replaced"] = int(dom_layerwise)
        # both roofs of the dominant launch, whichever binds: algorithmic (compulsory) bytes and measured (PMC) bytes over its
        # duration against the HBM peak; its folded FLOPs against the fp32 matrix peak
        roof["hbm"] = {"algorithmic_GBps": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
                       "traffic_GBps": round(traffic / (dom_ms * 1e-3) / 1e9, 1) if traffic else None,
                       "traffic_frac": round(traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                       "peak_GBps": HBM_PEAK_GBS}
        if dom_flops:
            tf_ = dom_flops / (dom_ms * 1e-3) / 1e12
            roof["mfma"] = {"TFLOPs": round(tf_, 2), "frac": round(tf_ / MFMA_F32_PEAK_TFLOPS, 4), "peak_TFLOPs": MFMA_F32_PEAK_TFLOPS}
        out = {
            "metric": "rendered Mtexels/s at %d^2 UV (full Model.call forward: U-Net + UV->camera warp)" % args.uv,
            "value": round(value, 2), "unit": "Mtexels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": PRECISION_DTYPE[args.precision],
            "data": "synthetic (seeded random uint8 texel buffers, %s uv2cam map, random-init weights of the released architecture)"
                    % ("chart-structured piecewise-smooth" if args.warp == 'charts' else "per-pixel uniform-random (adversarial)"),
            "config": {"workload": "BASELINE config 3: dragon_specular relight+view-synth, depth0 16/depth %d, "
                                   "%d frames/GPU, %d^2 UV, k=%d obs maps, %d^2 camera warp"
                                   % (args.depth, args.frames, args.uv, args.k, args.cam),
                       "frames_per_gpu": args.frames, "uv": args.uv, "k": args.k, "cam": args.cam,
                       "conv_algo": args.algo, "plan": "layer-by-layer" if args.no_fused else "fused ends",
                       "launch": "hipGraph replay" if args.graph else "eager, two HIP streams",
                       "distinct_batches_rotated": len(batches), "batch_source": "Dataset.load_batch (uint8 store -> float32 11-tuple, staging ring)",
                       "launch_tape_replays": fwd_replays, "parallelism": "dp%d (frames sharded, no forward collective)" % world,
                       "world": world, "dist_world_size": dist.get_world_size() if world > 1 else 1, "device": str(device)},
            "roofline": roof,
            "whole_pass": whole_pass(args, timer, elapsed / args.steps, bpt, live_step_bytes, plan=model.plan),
        }
        if with_loader:
            out["forward_including_loader"] = with_loader
        if world == 1 and not args.headline_only and args.uv == 1024:
            out["config5_2048_bf16"] = bench_config5(args, device)
            out["stress_64ch"] = bench_stress_64ch(device)
            if args.precision in ('fp32', 'f32x3_9') and not args.no_fused:
                others = bench_f32_split(args, device, model, batches, tuple(p_ for p_ in ('fp32', 'f32x3', 'f32x3_9') if p_ != args.precision))
                if 'fp32' in others:
                    out["native_fp32"] = others.pop('fp32')          # every product on v_mfma_f32: the round-1..3 headline
                out["config3_f32_split"] = others
        if released:
            out["released_shapes"] = released
        if world == 1 and not args.headline_only and args.uv == 1024 and not args.no_fused:
            out["nlt_test_infer"] = child_leg('--infer-mode-worker', extra=['--steps', str(args.steps), '--depth', str(args.depth)])
            if "error" in out["nlt_test_infer"]:
                try:
                    out["nlt_test_infer"] = bench_infer_mode(args, device)
                except Exception as e:                                # a sub-line: never take the line with it
                    out["nlt_test_infer"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        # The SAME metric at every N (review r05, weak point 8b): top-level `value` is the forward -- N replicas, no collective,
        # weak scaling by construction -- at N = 1, 2, 4, 8 alike, so value(N) / value(1) means something.  What shards WITH a
        # collective is BASELINE config 4's train step: `train_step.value` at every N, and at N > 1 its own scaling figures
        # (`speedup_vs_one_rank` = N x ms_no_collective / ms_overlapped, `scaling_efficiency` = that / N) measured in this run.
        out["metric"] += ("; at N > 1: N collective-free replicas of this forward (weak scaling).  The data-parallel TRAIN step with "
                          "the RCCL gradient all-reduce (BASELINE config 4) is `train_step`: value = Mtexels/s over all ranks, "
                          "speedup_vs_one_rank / scaling_efficiency at N > 1")
        if world > 1:
            out["config"]["rank_check"] = rank_check
            if train:
                out["config"]["train_step_parallelism"] = ("dp%d (frames sharded; gradient all-reduce over RCCL/xGMI in 3 fixed ranges, "
                                                           "ranges 0-1 issued inside the backward)" % world)
                out["config"]["scaling_curve"] = train[0]["comm"]["scaling_curve"]
            if backend != 'nccl' or os.environ.get('NLT_BENCH_SHARE_GPU', '0') == '1':
                out["config"]["dry_run"] = "backend %s, ranks share GPUs: a code-path check, NOT a measurement" % backend
        if train:
            out["train_step"] = train[0]
            if len(train) > 1:
                out["train_step_other_losses"] = train[1:]
        # several batches in flight: last of the GPU legs (the lanes' streams / graph pools stay with the process and were
        # measured to perturb legs that run after them: config 5 fp32 1.60 -> 1.70 ms)
        if world == 1 and not args.graph and (args.pipelined or not args.headline_only):
            out["pipelined"] = bench_pipelined(args, device, model, batches)
            if args.precision in ('fp32', 'f32x3_9') and not args.no_fused and not args.headline_only:
                try:
                    out["pipelined"]["f32x3_4_lanes"] = bench_f32_split_pipelined(args, device, model, batches)
                except Exception as e:                                # a sub-sub-line: never take the line with it
                    out["pipelined"]["f32x3_4_lanes"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            if released and args.uv == 1024:
                out["released_shapes"]["forward_4_frames_pipelined"] = released_pipelined_child(args)
        if world == 1 and not args.no_cpu_baseline and not args.headline_only:
            out["cpu_baseline"] = cpu_baseline(args)
        # what a 2000-byte tail of the line should still show goes last: the two baselines, then a digest of the line
        for key in ("native_fp32", "cpu_baseline"):
            if key in out:
                out[key] = out.pop(key)
        ts = {t_["loss"]: t_ for t_ in train if isinstance(t_, dict) and "loss" in t_}
        out["summary"] = {
            "forward_ms_per_step": out["ms_per_step"], "forward_Mtexels_per_s": out["value"], "forward_dtype": args.precision,
            "native_fp32_ms_per_step": out.get("native_fp32", {}).get("ms_per_step"),
            "dominant_kernel": dominant, "dominant_ms": round(dom_ms, 4), "dominant_frac_of_fp32_mfma_peak": roof.get("mfma", {}).get("frac"),
            "dominant_frac_of_hbm_peak_algorithmic": roof["hbm"]["frac"], "dominant_frac_of_hbm_peak_pmc_traffic": roof["hbm"]["traffic_frac"],
            "nlt_test_infer_1024_ms_per_step": out.get("nlt_test_infer", {}).get("uv1024_cam512", {}).get("ms_per_step"),
            "nlt_test_infer_1024_Mtexels_per_s": out.get("nlt_test_infer", {}).get("uv1024_cam512", {}).get("Mtexels_per_s"),
            "train_step_l2_ms": ts.get("l2", {}).get("ms_per_step"), "train_step_barron_ms": ts.get("barron", {}).get("ms_per_step"),
            "train_step_l2_host_enqueue_ms": ts.get("l2", {}).get("host_enqueue_ms_per_step"),
            "train_step_speedup_vs_one_rank": (train[0]["comm"].get("speedup_vs_one_rank") if train else None),
            "train_step_scaling_efficiency": (train[0]["comm"].get("scaling_efficiency") if train else None),
            "cpu_baseline_Mtexels_per_s": out.get("cpu_baseline", {}).get("value"), "cpu_cores": out.get("cpu_baseline", {}).get("cores")}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
