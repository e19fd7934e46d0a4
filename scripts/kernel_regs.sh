#!/bin/bash
# VGPR / SGPR / LDS / occupancy of the blend kernels as compiled (no GPU needed).
cd "$(dirname "$0")/.."
mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=on -Wno-unused-result -DNDEBUG -fno-slp-vectorize \
  -c /root/repo/free-surgs_amd/csrc/${1:-render}.hip --save-temps -o ${1:-render}.o 2>/dev/null
python3 - "${1:-render}" "${2:-blend}" <<'PY'
import re, subprocess, sys
s = open("/tmp/isa/%s-hip-amdgcn-amd-amdhsa-gfx950.s" % sys.argv[1]).read()
for chunk in s.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, chunk) or [None, "?"])[1]
    name = g("name")
    if sys.argv[2] not in name: continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0].replace("void ", "")
    v = int(g("vgpr_count")); alloc = -(-v // 8) * 8
    print("%-52s vgpr %3d (waves/SIMD %d) sgpr %3s lds %6s spill %s" % (dem[:52], v, min(8, 512 // alloc), g("sgpr_count"),
          g("group_segment_fixed_size"), g("vgpr_spill_count")))
PY
