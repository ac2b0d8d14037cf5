#!/bin/bash
# Developer tool: register / scratch / LDS figures of the step kernels of one build:  tools/kinfo.sh [extra hipcc flags]
cd "$(dirname "$0")/.."
mkdir -p build/kd
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden "$@" \
  --cuda-device-only -S -o build/kd/info.s nl-partsol_amd/csrc/nlps_gpu.hip 2>/dev/null
python3 - <<'PY'
import re
txt = open('build/kd/info.s').read()
for name in ["_Z7k2_tileILi3ELb1ELi256ELi1EE", "_Z7k3_tileILi3ELi0ELi1ELb0ELi256EE", "_Z7k3_tileILi3ELi1ELi1ELb0ELi256EE",
             "_Z7k3_tileILi3ELi2ELi1ELb0ELi256EE", "_Z7k5_tileILi3ELi0EE", "_Z8k_searchILi3EE"]:
    m = re.search(r"\.amdhsa_kernel (%s\S*)(.*?)\.end_amdhsa_kernel" % re.escape(name), txt, re.S)
    if not m:
        continue
    body = m.group(2)
    g = lambda k: re.search(r"\.amdhsa_%s (\S+)" % k, body).group(1)
    # instruction count
    s = txt.find("\n" + m.group(1) + ":")
    e = txt.find(".amdhsa_kernel " + m.group(1))
    n = sum(1 for l in txt[s:e].split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";")))
    print("%-40s vgpr %s scratch %s lds %s instr %d" % (name[:40], g("next_free_vgpr"), g("private_segment_fixed_size"),
                                                         g("group_segment_fixed_size"), n))
PY
