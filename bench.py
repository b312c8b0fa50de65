#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on its config[1]:
cutout cross-correlations/sec, 64x64 px cutout pairs, upsample=10, 1e5 pairs per GPU
resident in HBM, on N MI355X of one node (one process per GPU, weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (spx_xcorr_refine_f32: FFT cross-correlation ->
arg-max -> MFMA upsampled-DFT refine -> 5x5 quadratic fit) over the rank's whole batch,
followed -- when N > 1 -- by the RCCL gather of the per-cutout (dx, dy) to rank 0.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

PAIRS_PER_GPU = 100000          # BASELINE.json configs[1]
TILE = 64
UPSAMPLE = 10
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic HBM bytes per pair: two float32 tiles in, (dx,dy) float64 + int32 status out
BYTES_PER_PAIR = 2 * TILE * TILE * 4 + 16 + 4


# ---------------------------------------------------------------------------
# CPU baseline ("port"): the oracle timed on the host cores of this box, three legs.
#   pair_u10  -- THE SAME WORK as the GPU value: oracle.xcorr_refine(ref, img, upsample) on 64x64
#                float32 pairs (zero-padded cross-power spectrum, coarse arg-max, matrix-DFT window,
#                5x5 fit; float64 numpy -- the definition of the pair mode, not a tuned CPU code);
#   pair_u1   -- the reference's own composition for ONE cross-correlation: fftconvolve 'same' +
#                find_peak (cc.py:114 + cc.py:86), float32 FFT: an upper bound for any CPU code that
#                goes through scipy for the same pair;
#   reference -- cc.find_displacement restated (cc.py:21-95: 4 cross-correlations + interlace +
#                find_peak per call, NCC), counted as 4 cross-correlations per call.
# cpu_baseline.value is the FASTEST of the two legs that run the reference's own path (reference /
# pair_u1); the float64 definition leg is 16-45x slower and is reported under `legs` only.
# ---------------------------------------------------------------------------
def _cpu_worker(args):
    seed, n_u10, n_u1, n_ref, tile, ups = args
    import numpy as np
    import datagen
    from oracle import subpixal_oracle as orc
    ref, img, _ = datagen.pair_batch(seed, 8, tile)
    r5, im4, _ = datagen.dither_batch(seed, 8, tile)
    orc.xcorr_refine(ref[0], img[0], ups, full_grid=False)          # warm
    orc.pair_shift_u1(ref[0], img[0])
    orc.find_displacement(r5[0], *im4[0], cc_type='NCC')
    out = []
    t0 = time.perf_counter()
    for k in range(n_u10):
        orc.xcorr_refine(ref[k % 8], img[k % 8], ups, full_grid=False)
    out.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for k in range(n_u1):
        orc.pair_shift_u1(ref[k % 8], img[k % 8])
    out.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for k in range(n_ref):
        j = k % 8
        orc.find_displacement(r5[j], im4[j, 0], im4[j, 1], im4[j, 2], im4[j, 3], cc_type='NCC')
    out.append(time.perf_counter() - t0)
    return out


def kernel_name(tile, upsample, refine='default'):
    """Template instance the C-ABI dispatches to (spx_capi.hip): refinement-window blocks
    WB = 0 for upsample 1, else ceil((upsample + 5) / 16)."""
    wb = 0 if upsample == 1 else (upsample + 5 + 15) // 16
    if tile <= 32:
        return 'spx::pair32_kernel<%d, float>' % wb
    if tile <= 85:
        # the default refine arithmetic (float32) of the 64 tile; spx::RefineF64 is the SPX_REFINE_F64 form
        f64 = wb > 0 and refine == 'float64'          # the library's default on 33..85 px is float32
        arith = 'RefineF64' if f64 else 'RefineF32'
        return 'spx::pair_kernel<2, %d, 0, %s, float, spx::%s>' % (wb, 'true' if tile > 64 else 'false', arith)
    return 'spx::pair128_kernel<3, %d, 0, float>' % wb


def usable_cores():
    """Host threads this process may really use: the cgroup CPU quota (the GPU box
    shows all 256 hardware threads but grants a share), else affinity/cpu_count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(tile=TILE, upsample=UPSAMPLE, n_u10=1200, n_u1=3000, n_ref=2000):
    import multiprocessing as mp
    cores = usable_cores()
    if tile > 64:                      # larger cutouts cost ~4x per pair: keep the leg bounded
        n_u10, n_u1, n_ref = n_u10 // 4, n_u1 // 4, n_ref // 4
    ctx = mp.get_context('fork')
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(100 + i, n_u10, n_u1, n_ref, tile, upsample) for i in range(cores)])
    wall = time.perf_counter() - t0
    slow = [max(r[k] for r in res) for k in range(3)]     # slowest process per leg
    legs = {
        'reference_mode': {'value': 4.0 * cores * n_ref / slow[2], 'unit': 'cross-correlations/s',
                           'sample': 'oracle.find_displacement (cc.py:21-95 restated through scipy, NCC, float32 FFT): '
                                     '4 cross-correlations + interlace + find_peak per call x %d calls per process'
                                     % n_ref},
        'pair_u1': {'value': cores * n_u1 / slow[1], 'unit': 'cross-correlations/s',
                    'sample': "reference's own composition for one pair (fftconvolve 'same' + find_peak, "
                              "cc.py:114+86, float32 FFT) x %d per process" % n_u1},
        'pair_u%d_f64_definition' % upsample: {
            'value': cores * n_u10 / slow[0], 'unit': 'cross-correlations/s',
            'sample': 'oracle.xcorr_refine: the float64 numpy DEFINITION of the pair mode (zero-padded cross-power '
                      'spectrum, matrix-DFT window, 5x5 fit) x %d pairs per process; same work as the GPU value '
                      'but not a tuned CPU code: 16-45x slower than the legs above, do not read a speed-up off it'
                      % n_u10},
    }
    # cpu_baseline.value = the FASTEST of the two legs that run the reference's own CPU path (scipy
    # fftconvolve + find_peak, cc.py:114 + cc.py:86); the float64 definition leg is a sub-key only
    best = max(('reference_mode', 'pair_u1'), key=lambda k: legs[k]['value'])
    return {
        'value': legs[best]['value'],
        'unit': 'cross-correlations/s',
        'cores': cores,
        'kind': 'port',
        'value_leg': best,
        'sample': ("the reference's own CPU path restated (oracle, scipy float32 FFT), fastest of its two legs "
                   "= '%s': %s; %dx%d float32 cutouts, %d processes, rate = work / slowest process time; whole "
                   'CPU leg %.1f s wall' % (best, legs[best]['sample'], tile, tile, cores, wall)),
        'legs': legs,
    }


# ---------------------------------------------------------------------------
# Compute roofline (SURVEY 8d "record roofline.compute_fraction alongside the HBM fraction"):
# floating-point operations one pair executes against the 157.3 TFLOP/s f32 vector (= f32 MFMA)
# peak of MI355X_MICROARCH.md.  Nothing is hard-coded here: the figure is read from files under
# profiles/<round>/ that carry the fingerprint of the kernel sources they were made from, and is
# quoted only when that fingerprint is THIS build's:
#   sq_flops_<tile>_u<U>.json -- dynamic: SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F32 + SQ_INSTS_VALU_MFMA_MOPS_F32
#                                counter passes of this very command (tools/gpu_sq_flops.sh); preferred
#   kernel_flops.json         -- static: ISA census of the kernel instance (tools/kernel_flops.py --json),
#                                over-counts the branches a pair does not take; labelled "static"
# ---------------------------------------------------------------------------
F32_PEAK_TFLOPS = 157.3
PROFILE_ROUNDS = ('r03', 'r02', 'r01')


def kernel_family(tile, upsample):
    wb = 0 if upsample == 1 else (upsample + 5 + 15) // 16
    fam = '32' if tile <= 32 else '64' if tile <= 64 else '64fold' if tile <= 85 else '192'
    return '%s:%d' % (fam, wb)


def _profile_json(name):
    """First profiles/<round>/<name> whose `kernel_build` is this build's fingerprint."""
    build = kernel_build()
    for rnd in PROFILE_ROUNDS:
        try:
            d = json.load(open(os.path.join(ROOT, 'profiles', rnd, name)))
        except (OSError, ValueError):
            continue
        if d.get('kernel_build') == build:
            return d, 'profiles/%s/%s' % (rnd, name)
    return None, None


def flops_per_pair(tile, upsample, n_local):
    """(MFLOP per pair, split text, source) or None when no file matches this build."""
    d, src = _profile_json('sq_flops_%d_u%d.json' % (tile, upsample))
    if d is not None and d.get('pairs_per_launch') == n_local:
        return (d['flop_per_pair'] / 1e6,
                'dynamic, SQ counters: vector %.3f + matrix %.3f MFLOP per pair' % (
                    d['vector_flop_per_pair'] / 1e6, d['matrix_flop_per_pair'] / 1e6), src)
    d, src = _profile_json('kernel_flops.json')
    if d is not None:
        k = d['kernels'].get(kernel_family(tile, upsample))
        if k is not None:
            return (k['vector_mflop'] + k['matrix_mflop'],
                    'static ISA census (over-counts un-taken branches): vector %.3f + matrix %.3f MFLOP per pair'
                    % (k['vector_mflop'], k['matrix_mflop']), src)
    return None


def kernel_build():
    """Fingerprint of the kernel sources (ties a PMC traffic file to the build it was measured on;
    works on the GPU box, where there is no .git).  Comments and white space do not count: the hash
    is over the code the compiler sees."""
    import hashlib
    import re
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'subpixal_amd', 'csrc')
    for name in sorted(os.listdir(csrc)):
        if name.endswith(('.h', '.hip')):
            text = open(os.path.join(csrc, name), encoding='utf-8').read()
            text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)       # block comments
            text = re.sub(r'//[^\n]*', ' ', text)                     # line comments (no "//" inside the few string literals)
            h.update(name.encode())
            h.update(' '.join(text.split()).encode())
    return h.hexdigest()[:16]


REF_SOURCES = 20000             # reference-mode block: sources per launch (1.6 GB of cutouts)


def reference_mode_block(dev, steps, warmup):
    """cc.find_displacement (cc.py:21-95) -- the function subpixal itself calls (align.py:682-685) -- for a batch
    of REF_SOURCES 64x64 float32 sources with their four half-pixel dithers, NCC, inputs resident in HBM:
    displacements/s from HIP events on the launch stream.  Reported beside the headline, never in `value`."""
    import torch
    import subpixal_amd
    n, N = TILE, REF_SOURCES
    g = torch.Generator(device=dev)
    g.manual_seed(20261004)
    u = torch.rand((N, 4), generator=g, device=dev, dtype=torch.float64)
    tx, ty = (2 * u[:, 0] - 1) * 3.0, (2 * u[:, 1] - 1) * 3.0
    sig, amp = 4.0 + 2.0 * u[:, 2], 0.5 + 1.5 * u[:, 3]
    yy, xx = torch.meshgrid(torch.arange(n, device=dev, dtype=torch.float64),
                            torch.arange(n, device=dev, dtype=torch.float64), indexing='ij')
    c = (n - 1) / 2.0

    def spots(x0, y0):
        r2 = (xx[None] - x0[:, None, None]) ** 2 + (yy[None] - y0[:, None, None]) ** 2
        return (amp[:, None, None] * torch.exp(-r2 / (2.0 * sig[:, None, None] ** 2))).to(torch.float32)
    ref = spots(torch.full_like(tx, c), torch.full_like(ty, c))
    im4 = torch.empty((N, 4, n, n), dtype=torch.float32, device=dev)
    # dither convention of align.py:664-676: image10(x, y) = image00(x + 1/2, y)
    for q, (ox, oy) in enumerate(((0.0, 0.0), (0.5, 0.0), (0.0, 0.5), (0.5, 0.5))):
        im4[:, q] = spots(c + tx - ox, c + ty - oy)
    truth = torch.stack([tx, ty], dim=1)
    for _ in range(max(1, warmup)):
        d = subpixal_amd.find_displacement_batch(ref, im4, cc_type='NCC')
    torch.cuda.synchronize()
    err = float((d - truth).abs().max())
    # (the reference's own 5x5 fit is biased by up to 7e-4 px at sigma = 4 px: SURVEY.md 8 a-0)
    assert err < 2e-3, "reference mode is wrong (%g px): refusing to time" % err
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record()
        subpixal_amd.find_displacement_batch(ref, im4, cc_type='NCC')
        b.record()
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in ev) / steps
    bytes_per = 5 * n * n * 4 + 16 + 4
    rate = N / (ms * 1e-3)
    pmc, src = _profile_json('pmc_traffic_disp5_64.json')
    traffic = pmc['hbm_bytes_per_launch'] if pmc is not None and pmc.get('sources_per_launch') == N else None
    # (the traffic includes the interlaced image, 16 n^2 bytes per source, written on request of the caller --
    #  align.py:684 always asks; bytes_per_displacement does not)
    return {
        'what': 'cc.find_displacement (cc.py:21-95) for a batch: 5 cutouts per source, NCC, interlaced image '
                'written to HBM; same process, after the headline loop; NOT part of `value`',
        'value': rate, 'unit': 'displacements/s', 'cross_correlations_per_s': 4.0 * rate,
        'sources_per_launch': N, 'cutout': n, 'cc_type': 'NCC', 'steps': steps,
        'kernel': 'spx::p5::disp5p_kernel<false, float>', 'kernel_ms': ms,
        'bytes_per_displacement': bytes_per,
        'achieved_GBps': rate * bytes_per / 1e9, 'frac_of_hbm_peak': rate * bytes_per / 1e9 / HBM_PEAK_GBS,
        'traffic': traffic, 'traffic_source': src if traffic is not None else None,
        'max_abs_err_px_vs_truth': err,
    }


def launch_ranks(n, argv):
    """`python bench.py --gpus N` as typed: one child `python -m torch.distributed.run` with N ranks on
    127.0.0.1 (a free port), stdout/stderr passed through.  Returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--pairs', type=int, default=None,
                    help='pairs per GPU (default 1e5 = configs[1]; at 8 GPUs 1.25e6 = configs[3], 1e7 pairs in all)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-reference-mode', action='store_true',
                    help='skip the reference-mode (cc.find_displacement) block of the N=1 line')
    ap.add_argument('--tile', type=int, default=TILE, help='cutout side (64 = config 2, 128 = config 3)')
    ap.add_argument('--upsample', type=int, default=UPSAMPLE)
    ap.add_argument('--refine', default='default', choices=['default', 'float64', 'float32'],
                    help="arithmetic of the refine stage (SPX_REFINE_*): 'float64' = the slower, more precise form of the "
                         "64 tile / fold path; the graded line is the default")
    ap.add_argument('--backend', default='nccl', help="'gloo' + --one-device rehearses the N>1 path on one GPU")
    ap.add_argument('--one-device', action='store_true', help='every rank uses cuda:0 (rehearsal only)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks as CHILD processes (one per GPU, the same
        # command the driver uses) and relay rank 0's JSON line.  This process has not touched the GPU
        # and never does; it only waits and exits with the children's code.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        args.gpus = world

    # CPU baseline first (fork pool), before this process touches the GPU
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.tile, args.upsample)

    import torch
    import torch.distributed as dist
    import subpixal_amd
    from subpixal_amd import synth, dist as spx_dist

    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)
    on_cpu = world > 1 and args.backend != 'nccl'      # gloo rehearsal: collectives on host copies

    def gather(d, async_op=False):
        if world == 1:
            return spx_dist.PendingGather(None, None, None, None, local=d) if async_op else d
        return spx_dist.gather_shifts(d.cpu() if on_cpu else d, n_total=n_total, dst=0, async_op=async_op)

    # BASELINE.json configs[3]: 1e7 64x64 pairs over 8 GPUs = 1.25e6 pairs (41 GB) per GPU
    config4 = args.pairs is None and world == 8 and args.tile == TILE
    n_local = args.pairs if args.pairs is not None else (1250000 if config4 else PAIRS_PER_GPU)
    n_total = n_local * world
    # inputs generated on the device, resident in HBM before the timed region
    tile, ups = args.tile, args.upsample
    bytes_per_pair = 2 * tile * tile * 4 + 16 + 4
    # sigma ~ U(4,6) (n = 32: U(3,4), and shifts within +-2 px so the spot stays
    # inside the small tile)
    gen = dict(sigma_lo=3.0, sigma_hi=4.0, max_shift=2.0) if tile <= 32 else {}
    ref, img, truth = synth.gaussian_pairs(n_local, tile, seed=20261003,
                                           first_index=rank * n_local, dev=local_rank, **gen)
    torch.cuda.synchronize()

    def step():
        d = subpixal_amd.xcorr_refine_batch(ref, img, upsample=ups, refine=args.refine)
        g = gather(d)
        return d, g

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        d, g = step()
    torch.cuda.synchronize()
    # sanity: the thing being timed is the correct answer
    err = float((d - truth).abs().max())
    # (n = 32 spots are clipped by the tile, which biases the truth comparison itself)
    assert err < (1e-3 if tile > 32 else 1e-2), "shifts are wrong (%g px): refusing to time" % err

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # every step's shifts are gathered on rank 0 (north_star: "at most an RCCL gather of the per-cutout
    # (dx, dy) shifts"); one gather stays in flight on RCCL's stream while the next step's kernel runs
    pending = None
    for k in range(args.steps):
        ev[k][0].record()                 # torch's current stream == the launch stream
        d = subpixal_amd.xcorr_refine_batch(ref, img, upsample=ups, refine=args.refine)
        ev[k][1].record()
        if pending is not None:
            pending.result()
        pending = gather(d, async_op=True)
    pending.result()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if on_cpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    refmode = None
    if rank == 0 and world == 1 and not args.no_reference_mode and tile == TILE:
        del ref, img
        refmode = reference_mode_block(dev, max(3, args.steps // 5), 2)

    if rank == 0:
        # HBM traffic of the same launch from the committed PMC passes (rocprofv3 cannot run
        # inside this process); only quoted for the configuration it was measured on
        # ... and only when that file was measured on THIS build of the kernels
        name = 'pmc_traffic.json' if (tile == TILE and ups == UPSAMPLE) else 'pmc_traffic_%d_u%d.json' % (tile, ups)
        traffic, traffic_src = None, None
        pmc, src = _profile_json(name)
        # (the counter files were taken on the default refine form: a --refine float64 line quotes none of them)
        counters_apply = args.refine == 'default' or tile <= 32 or tile > 85     # (measured with the default)
        if counters_apply and pmc is not None and pmc.get('pairs_per_launch') == n_local and \
                pmc.get('tile', TILE) == tile and pmc.get('upsample', UPSAMPLE) == ups:
            traffic, traffic_src = pmc['hbm_bytes_per_launch'], src
        value = n_total * args.steps / elapsed
        achieved = n_local * bytes_per_pair / (kern_ms * 1e-3) / 1e9
        out = {
            'metric': 'cutout cross-correlations/sec (%dx%d px, upsample=%d)' % (tile, tile, ups),
            'value': value,
            'unit': 'cross-correlations/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {
                'workload': 'BASELINE.json configs[%d]%s: %d %dx%d Gaussian-spot cutout pairs per GPU, '
                            'upsample=%d, inputs resident in HBM'
                            % (3 if config4 else 0 if tile <= 32 else 1 if tile <= 64 else 2,
                               ' (1e7 pairs over 8 GPUs)' if config4 else
                               ' shape, on the GPU' if tile <= 32 else ('' if tile in (64, 128) else ' family'),
                               n_local, tile, tile, ups),
                'pairs_per_gpu': n_local, 'tile': tile, 'upsample': ups, 'cc_type': 'CC', 'refine': args.refine,
                'parallelism': 'batch sharded over %d GPU(s); gather of (dx,dy) to rank 0' % world,
            },
            'roofline': {
                'bound': 'hbm',
                'achieved': achieved,
                'peak': HBM_PEAK_GBS,
                'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS,
                'traffic': traffic,
                'traffic_unit': 'HBM bytes per launch from the PMC passes of the same command on this kernel '
                                'build (%s); null = not measured for this build/config' % traffic_src,
                'algorithmic_bytes_per_launch': n_local * bytes_per_pair,
                'kernel': kernel_name(tile, args.upsample, args.refine),
                'kernel_ms': kern_ms,
                'bytes_per_pair': bytes_per_pair,
                'pairs_per_launch': n_local,
            },
        }
        fl = flops_per_pair(tile, ups, n_local) if counters_apply else None
        if fl is not None:
            tf = fl[0] * 1e6 * n_local / (kern_ms * 1e-3) / 1e12
            out['roofline'].update({
                'compute_fraction': tf / F32_PEAK_TFLOPS,
                'compute_achieved_tflops': tf,
                'compute_peak_tflops': F32_PEAK_TFLOPS,
                'flops_per_pair': fl[0] * 1e6,
                'flops_split': fl[1],
                'flops_source': fl[2],
            })
        out.update({
            'max_abs_err_px_vs_truth': err,
        })
        if refmode is not None:
            out['reference_mode'] = refmode
        if cpu is not None:
            out['cpu_baseline'] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
