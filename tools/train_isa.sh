#!/bin/bash
# hipcc -S of csrc/photo_train.hip with the shipped flags (+ extra -D...) and a resource / loop report (no GPU needed):
#   bash tools/train_isa.sh out.s [-DMDX_...=...]
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall \
    -Wno-unused-function -fno-slp-vectorize -S --cuda-device-only "$@" -o "$OUT" \
    "$ROOT/digging-into-self-supervised-monocular-depth-estimation_amd/csrc/photo_train.hip" 2>&1 | grep -v "hip-link" | head -30
python3 - "$OUT" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", t, re.S):
    b = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    if "ELb1ELb1" in name or "Li2ELb0ELb1" in name:
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1))   # noqa: E731
        print(name[24:46], "vgpr", g("vgpr_count"), "spill", g("vgpr_spill_count"), "sgpr_spill", g("sgpr_spill_count"),
              "lds", g("group_segment_fixed_size"), "scratch", g("private_segment_fixed_size"))
PY
for k in Li2ELb1ELb1 Li3ELb1ELb1; do
    python3 "$ROOT/tools/isa_spills.py" "$OUT" $k | tail -1
    python3 "$ROOT/tools/isa_loop.py" "$OUT" $k
done
