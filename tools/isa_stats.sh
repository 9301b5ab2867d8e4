#!/bin/bash
# Per-kernel ISA statistics of the HIP sources (CPU box: hipcc cross-compiles gfx950).  DESIGN.md 5f / 7c:
#   registers, scratch, occupancy, spilled scalars (v_writelane / v_readlane), LDS permutes
#   vector loads against FULL waits (s_waitcnt vmcnt(0)) and counted waits, scalar loads, stores, atomics
# A kernel whose full waits approach its load count runs its loads one round trip at a time (a load / store / atomic behind a
# per-element branch: DESIGN.md 5f).
#   bash tools/isa_stats.sh [file.hip ...] [-k kernel-name-regex]
set -e
cd "$(dirname "$0")/.."
PAT="."
FILES=()
while [ $# -gt 0 ]; do
  if [ "$1" = "-k" ]; then PAT="$2"; shift 2; else FILES+=("$1"); shift; fi
done
[ ${#FILES[@]} -eq 0 ] && FILES=(prodsearch_amd/csrc/*.hip)
OUT=$(mktemp -d)
for f in "${FILES[@]}"; do
  s="$OUT/$(basename "$f" .hip).s"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o "$s" "$f" 2>/dev/null
  awk -v pat="$PAT" '
    /^_Z[A-Za-z0-9_]*:/{name=$1; sub(":","",name)}
    /^\t(global|buffer)_load/{gl[name]++} /s_waitcnt.*vmcnt\(0\)/{v0[name]++} /s_waitcnt.*vmcnt\([1-9]/{vn[name]++}
    /^\ts_load/{sl[name]++} /^\t[a-z]/{n[name]++} /global_atomic/{at[name]++} /^\t(global|buffer)_store/{st[name]++}
    /v_readlane/{rl[name]++} /v_writelane/{wl[name]++} /ds_bpermute/{bp[name]++}
    /; NumVgprs:/{v[name]=$3} /; ScratchSize:/{ss[name]=$3} /; Occupancy:/{oc[name]=$3}
    END{for(k in v) if (k ~ pat) printf "%-60s vgpr=%3s scratch=%3s occ=%s | loads=%3d full-waits=%3d counted=%3d sloads=%3d stores=%3d atomics=%3d | writelane=%d readlane=%d bpermute=%d\n",
         substr(k,1,60), v[k], ss[k], oc[k], gl[k], v0[k], vn[k], sl[k], st[k], at[k], wl[k], rl[k], bp[k]}' "$s" | sort
done
rm -rf "$OUT"
