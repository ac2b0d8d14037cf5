#!/bin/bash
# Developer tool: register / scratch / LDS figures of the tile kernels of one build:  tools/kinfo.sh [extra hipcc flags]
# (every k2_tile / k3_tile / k3_tile_lazy / k5_tile / k5_tile_lazy instantiation of the 3-D build, demangled)
cd "$(dirname "$0")/.."
mkdir -p build/kd
[ -n "$KINFO_REUSE" ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden "$@" \
  --cuda-device-only -S -o build/kd/info.s nl-partsol_amd/csrc/nlps_gpu.hip 2>/dev/null
python3 - <<'PY'
import re, subprocess
txt = open('build/kd/info.s').read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    if not re.match(r"_Z\d+k[235]_tile", name):
        continue
    dem = subprocess.run(["/usr/bin/c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.split("(")[0].replace("void ", "")
    if "<3," not in dem:
        continue
    g = lambda k: re.search(r"\.amdhsa_%s (\S+)" % k, body).group(1)
    s = txt.find("\n" + name + ":")
    e = txt.find(".amdhsa_kernel " + name)
    n = sum(1 for l in txt[s:e].split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";")))
    print("%-44s vgpr %3s agpr %3s scratch %4s lds %6s instr %5d" % (dem[:44], g("next_free_vgpr"), g("accum_offset") if False else "-",
                                                         g("private_segment_fixed_size"), g("group_segment_fixed_size"), n))
PY
