#!/usr/bin/env python3
"""Print VGPR/AGPR/SGPR counts, spills, scratch and LDS of every kernel in a hipcc --save-temps
.s file (or build one from spx_capi.hip into /tmp/isa)."""
import re
import subprocess
import sys

path = sys.argv[1] if len(sys.argv) > 1 else None
if path is None:
    import os
    os.makedirs('/tmp/isa', exist_ok=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
                           '-Wno-unused-function', '--save-temps', '-o', '/tmp/isa/lib.so',
                           os.path.join(root, 'subpixal_amd/csrc/spx_capi.hip')], cwd='/tmp/isa',
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    path = '/tmp/isa/spx_capi-hip-amdgcn-amd-amdhsa-gfx950.s'
txt = open(path).read()
meta = txt[txt.index('amdhsa.kernels:'):]
for blk in meta.split('  - .agpr_count:')[1:]:
    g = lambda k: re.search(r'\.' + k + r':\s*(\S+)', blk)
    name = subprocess.run(['c++filt', g('name').group(1)], capture_output=True, text=True).stdout.strip()
    name = name.split('(')[0].replace('void ', '')
    agpr = blk.split('\n')[0].strip()
    print(f"{name:42s} vgpr {g('vgpr_count').group(1):>4s} agpr {agpr:>3s} sgpr {g('sgpr_count').group(1):>4s} "
          f"spill {g('vgpr_spill_count').group(1):>4s} scratch {g('private_segment_fixed_size').group(1):>5s} "
          f"lds {g('group_segment_fixed_size').group(1):>6s}")
