"""CPU-side checks of the boundary: the C-ABI library loads and exports every
symbol include/subpixal_hip.h declares; host-side argument/error conventions;
multi-rank sharding/gather under gloo."""
import ctypes
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'subpixal_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(spx_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from subpixal_amd import _ffi
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_ffi.EXPORTED_SYMBOLS) == names
    assert lib.spx_abi_version() == _ffi.ABI_VERSION == 4


def test_argument_errors_without_gpu():
    from subpixal_amd import _ffi
    lib = _ffi.load()
    assert lib.spx_workspace_bytes_displacement5(10, 64, 64, 1) == 10 * 4 * 64 * 64 * 4
    assert lib.spx_workspace_bytes_displacement5(10, 64, 64, 0) == 0
    assert lib.spx_workspace_bytes_xcorr(10, 64, 64) == 0
    # cutouts up to 85 px stay in LDS (64 tile's fold path, period 128); above that the period-192
    # path: 9 complex class planes of 64x64 + the 192 x (192+4) convolution, per workgroup
    assert lib.spx_workspace_bytes_xcorr(10, 85, 64) == 0
    assert lib.spx_workspace_bytes_xcorr(10, 86, 64) == 10 * (9 * 2 * 64 * 64 + 192 * 196) * 4
    assert lib.spx_workspace_bytes_xcorr(10, 97, 128) == 10 * (9 * 2 * 64 * 64 + 192 * 196) * 4
    assert lib.spx_workspace_bytes_displacement5(10, 86, 64, 0) == 10 * (9 * 2 * 64 * 64 + 192 * 196) * 4
    # argument validation happens before any HIP call
    assert lib.spx_xcorr_refine_f32(None, None, 1, 64, 64, 1, 0, None, None, None, 0, None) == -1
    assert lib.spx_xcorr_refine_f32(None, None, 0, 64, 64, 1, 0, None, None, None, 0, None) == 0
    assert lib.spx_xcorr_refine_f32(None, None, -1, 64, 64, 1, 0, None, None, None, 0, None) == -1
    buf = (ctypes.c_double * 4)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.spx_xcorr_refine_f32(p, p, 1, 683, 64, 1, 0, p, None, None, 0, None) == -2
    assert b'5..682' in lib.spx_last_error()
    # the general path (129..682 px): 4 C^2 planes of 64x64 + the P x (P+4) convolution per workgroup
    assert lib.spx_workspace_bytes_xcorr(3, 129, 64) == 3 * (4 * 16 * 64 * 64 + 256 * 260) * 4
    assert lib.spx_workspace_bytes_xcorr(3, 682, 64) == 3 * (4 * 256 * 64 * 64 + 1024 * 1028) * 4
    assert lib.spx_xcorr_refine_f32(p, p, 1, 200, 64, 1, 0, p, None, None, 0, None) == -4
    assert lib.spx_xcorr_refine_f32(p, p, 1, 86, 64, 1, 0, p, None, None, 0, None) == -4     # period 192 needs workspace
    assert lib.spx_xcorr_refine_f64(p, p, 1, 86, 64, 1, 0, p, None, None, 0, None) == -4
    assert lib.spx_xcorr_refine_f64(None, None, 1, 64, 64, 1, 0, None, None, None, 0, None) == -1
    assert lib.spx_find_displacement5_f64(p, p, 1, 128, 128, 0, p, None, p, None, 0, None) == -4
    assert lib.spx_shutdown() == 0           # nothing initialised: nothing to free
    assert lib.spx_xcorr_refine_f32(p, p, 1, 64, 64, 60, 0, p, None, None, 0, None) == -2
    assert lib.spx_find_displacement5_f32(p, p, 1, 128, 128, 0, p, None, p, None, 0, None) == -4
    assert lib.spx_find_peak_f64(p, None, None, 1, 8, 8, 0, 5, 0, 0, p, None, None) == -1
    assert lib.spx_xcorr_refine_f32(p, p, 0, 64, 64, 1, 0, p, None, None, 0, None) == 0   # empty batch
    # the same entry with the refine arithmetic chosen per call (SPX_REFINE_DEFAULT = 0, SPX_REFINE_F64 = 1, SPX_REFINE_F32 = 2)
    for fn in (lib.spx_xcorr_refine_ex_f32, lib.spx_xcorr_refine_ex_f64):
        assert fn(p, p, 0, 64, 64, 10, 0, 0, p, None, None, 0, None) == 0               # empty batch, default
        assert fn(p, p, 0, 64, 64, 10, 0, 1, p, None, None, 0, None) == 0               # empty batch, float64 refine
        assert fn(p, p, 0, 64, 64, 10, 0, 2, p, None, None, 0, None) == 0               # empty batch, float32 refine
        assert fn(p, p, 0, 64, 64, 10, 0, 3, p, None, None, 0, None) == -1              # no such arithmetic
        assert fn(p, p, 0, 64, 64, 10, 0, -1, p, None, None, 0, None) == -1
        assert fn(None, None, 1, 64, 64, 10, 0, 1, None, None, None, 0, None) == -1     # null pointers
        assert fn(p, p, 1, 683, 64, 10, 0, 1, p, None, None, 0, None) == -2             # shape
        assert fn(p, p, 1, 86, 64, 10, 0, 1, p, None, None, 0, None) == -4              # period 192 needs workspace


def test_python_api_errors_mirror_reference():
    import subpixal_amd
    from subpixal_amd import centroid
    a = np.zeros((8, 8), np.float32)
    with pytest.raises(ValueError, match="All cutouts must have same shape."):
        subpixal_amd.find_displacement(a, a, a, np.zeros((8, 9), np.float32), a)
    with pytest.raises(ValueError, match="Both 'xmax' and 'ymax'"):
        subpixal_amd.find_peak(a, xmax=1.0)
    with pytest.raises(TypeError):
        centroid._process_box_pars((1, 2, 3))
    with pytest.raises(ValueError):
        centroid._process_box_pars(0)
    # no GPU here: the product path must fail loudly, not fall back
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            subpixal_amd.find_displacement(a, a, a, a, a)


def test_py2round():
    from subpixal_amd.utils import py2round
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'find_peak.npz'))
    np.testing.assert_array_equal(py2round(g['py2round_x']), g['py2round_array'])
    for v, e in zip(g['py2round_x'], g['py2round_scalar']):
        assert float(py2round(float(v))) == e


def test_shard_range():
    from subpixal_amd.dist import shard_range
    for n in (0, 1, 7, 100000, 10**7 + 3):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_gather_without_a_process_group_is_the_identity():
    """one process (bench.py --gpus 1, the python API): nothing to exchange, sync or async"""
    import torch
    from subpixal_amd.dist import gather_shifts, PendingGather
    local = torch.arange(10, dtype=torch.float64).reshape(5, 2)
    assert gather_shifts(local) is local
    pend = gather_shifts(local, async_op=True)
    assert isinstance(pend, PendingGather) and pend.result() is local and pend.result() is local


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
import numpy as np, torch, torch.distributed as dist
from subpixal_amd.dist import xcorr_refine_sharded, shard_range
import datagen
from oracle import subpixal_oracle as orc
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
N = 9
tx, ty, sg, am = datagen.random_params(5, N, 64)
def make(lo, hi):
    r = np.stack([datagen.pair_set(64, 64, tx[k], ty[k], sg[k], am[k])[0] for k in range(lo, hi)])
    i = np.stack([datagen.pair_set(64, 64, tx[k], ty[k], sg[k], am[k])[1] for k in range(lo, hi)])
    return r, i
def compute(ref, img, upsample, cc_type):      # stand-in for the GPU kernel under gloo
    return torch.from_numpy(orc.xcorr_refine_batch(ref, img, upsample, cc_type)[0])
out = xcorr_refine_sharded(make, N, upsample=1, compute=compute)
# the pipelined form bench.py uses: one gather in flight while the next block is computed
from subpixal_amd.dist import gather_shifts
lo, hi = shard_range(N, rank, world)
pend, outs = None, []
for rep in range(3):
    local = compute(*make(lo, hi), 1, 'CC') + float(rep)
    if pend is not None:
        outs.append(pend.result())
    pend = gather_shifts(local, n_total=N, dst=0, async_op=True)
outs.append(pend.result())
if rank == 0:
    full = orc.xcorr_refine_batch(*make(0, N), 1)[0]
    assert out.shape == (N, 2)
    assert np.array_equal(out.numpy(), full)
    for rep, o in enumerate(outs):
        assert np.array_equal(o.numpy(), full + float(rep))
    print('GLOO_OK')
else:
    assert out is None and all(o is None for o in outs)
dist.destroy_process_group()
'''


def test_sharded_driver_gloo_world2(tmp_path):
    script = tmp_path / 'w.py'
    script.write_text(_WORKER)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    res = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                          '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert 'GLOO_OK' in res.stdout


def test_header_compiles_as_plain_c(tmp_path):
    """include/subpixal_hip.h is the drop-in boundary: it must be usable from C (the language a
    cgo / JNI / ctypes-free binding would consume), not only from C++."""
    import re
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, 'include', 'subpixal_hip.h')).read()
    names = sorted(set(re.findall(r'\b(spx_[a-z0-9_]+)\s*\(', header)))
    assert len(names) >= 14
    src = tmp_path / 'use_abi.c'
    src.write_text('#include "subpixal_hip.h"\n'
                   'const void* table[] = {\n' + ''.join('    (const void*)%s,\n' % n for n in names) + '};\n'
                   'int abi(void) { return SPX_ABI_VERSION + SPX_MAX_SIDE + SPX_E_WORKSPACE + SPX_ST_FEWPTS; }\n')
    subprocess.check_call([gcc, '-std=c99', '-Wall', '-Wextra', '-Werror', '-pedantic', '-Wno-pedantic',
                           '-I', os.path.join(root, 'include'), '-c', str(src), '-o', str(tmp_path / 'use_abi.o')])
