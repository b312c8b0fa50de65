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
# CPU baseline ("port"): the oracle's restatement of the reference's own per-source
# path cc.find_displacement (cc.py:21-95: 4 fftconvolve cross-correlations of 64x64
# float32 cutouts + interlace + find_peak), timed on the host cores of this box.
# ---------------------------------------------------------------------------
def _cpu_worker(args):
    seed, count = args
    import numpy as np
    import datagen
    from oracle import subpixal_oracle as orc
    ref, im4, _ = datagen.dither_batch(seed, 8, TILE)
    orc.find_displacement(ref[0], *im4[0], cc_type='NCC')          # warm
    t0 = time.perf_counter()
    for k in range(count):
        j = k % 8
        orc.find_displacement(ref[j], im4[j, 0], im4[j, 1], im4[j, 2], im4[j, 3], cc_type='NCC')
    return count, time.perf_counter() - t0


def kernel_name(tile, upsample):
    """Template instance the C-ABI dispatches to (spx_capi.hip): refinement-window blocks
    WB = 0 for upsample 1, else ceil((upsample + 5) / 16)."""
    wb = 0 if upsample == 1 else (upsample + 5 + 15) // 16
    if tile <= 32:
        return 'spx::pair32_kernel<%d>' % wb
    if tile <= 64:
        return 'spx::pair_kernel<2,%d>' % wb
    return 'spx::pair128_kernel<%d,%d>' % (3 if tile <= 96 else 4, wb)


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


def cpu_baseline(per_core=2500):
    import multiprocessing as mp
    cores = usable_cores()
    ctx = mp.get_context('fork')
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(100 + i, per_core) for i in range(cores)])
    wall = time.perf_counter() - t0
    busy = max(r[1] for r in res)
    total = sum(r[0] for r in res)
    return {
        'value': 4.0 * total / busy,
        'unit': 'cross-correlations/s',
        'cores': cores,
        'kind': 'port',
        'sample': ('oracle.find_displacement (reference cc.py:21-95 restated: 4 cross-correlations '
                   '+ 2x interlace + find_peak per call, NCC, 64x64 float32) x %d calls on each of '
                   '%d processes; rate = 4*calls / slowest process time (%.1f s, wall %.1f s)'
                   % (per_core, cores, busy, wall)),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--pairs', type=int, default=PAIRS_PER_GPU)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--tile', type=int, default=TILE, help='cutout side (64 = config 2, 128 = config 3)')
    ap.add_argument('--upsample', type=int, default=UPSAMPLE)
    ap.add_argument('--backend', default='nccl', help="'gloo' + --one-device rehearses the N>1 path on one GPU")
    ap.add_argument('--one-device', action='store_true', help='every rank uses cuda:0 (rehearsal only)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
        args.gpus = world

    # CPU baseline first (fork pool), before this process touches the GPU
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

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

    def gather(d):
        if world == 1:
            return d
        return spx_dist.gather_shifts(d.cpu() if on_cpu else d, n_total=n_total, dst=0)

    n_local = args.pairs
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
        d = subpixal_amd.xcorr_refine_batch(ref, img, upsample=ups)
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
    for k in range(args.steps):
        ev[k][0].record()                 # torch's current stream == the launch stream
        d = subpixal_amd.xcorr_refine_batch(ref, img, upsample=ups)
        ev[k][1].record()
        gather(d)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if on_cpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps

    if rank == 0:
        # HBM traffic of the same launch from the committed PMC passes (rocprofv3 cannot run
        # inside this process); only quoted for the configuration it was measured on
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01', 'pmc_traffic.json')))
            if pmc.get('pairs_per_launch') == n_local and tile == TILE and ups == UPSAMPLE:
                traffic = pmc['hbm_bytes_per_launch']
        except (OSError, ValueError, KeyError):
            pass
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
                            % (0 if tile <= 32 else 1 if tile <= 64 else 2,
                               ' shape, on the GPU' if tile <= 32 else (' family, 96 tile' if 64 < tile <= 96 else ''),
                               n_local, tile, tile, ups),
                'pairs_per_gpu': n_local, 'tile': tile, 'upsample': ups, 'cc_type': 'CC',
                'parallelism': 'batch sharded over %d GPU(s); gather of (dx,dy) to rank 0' % world,
            },
            'roofline': {
                'bound': 'hbm',
                'achieved': achieved,
                'peak': HBM_PEAK_GBS,
                'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS,
                'traffic': traffic,
                'traffic_unit': 'bytes per launch (2*FETCH_SIZE + WRITE_SIZE, profiles/r01/pmc_traffic.json)',
                'algorithmic_bytes_per_launch': n_local * bytes_per_pair,
                'kernel': kernel_name(tile, args.upsample),
                'kernel_ms': kern_ms,
                'bytes_per_pair': bytes_per_pair,
                'pairs_per_launch': n_local,
            },
            'max_abs_err_px_vs_truth': err,
        }
        if cpu is not None:
            out['cpu_baseline'] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
