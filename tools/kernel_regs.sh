#!/bin/bash
# Registers / scratch / LDS of every kernel in the given .hip files (gfx950 ISA metadata): name, vgprs, agprs, sgprs, scratch bytes.
# usage: tools/kernel_regs.sh diffusion_model_amd/csrc/edge_x_m16.hip [more.hip ...]   (extra hipcc flags via EXTRA=...)
set -e
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize $EXTRA -S --cuda-device-only "$f" -o /tmp/kregs.s 2>/dev/null
  python3 - "$f" <<'PY'
import re, sys
txt = open('/tmp/kregs.s').read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
    name, body = m.group(1), m.group(2)
    def g(k):
        r = re.search(r'\.amdhsa_%s (\S+)' % k, body)
        return r.group(1) if r else '?'
    import subprocess
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r'\(egnn::.*', '', dem)[:110]
    print(f"{dem:110s} vgpr {g('next_free_vgpr'):>4s} accum_off {g('accum_offset'):>4s} sgpr {g('next_free_sgpr'):>4s} scratch {g('private_segment_fixed_size'):>5s}")
PY
done
