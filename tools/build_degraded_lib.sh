#!/usr/bin/env bash
# A deliberately degraded library for the "the gates turn red" check: the register-stationary LSTM step rounds its
# accumulators to bf16 after every 32-k MFMA step (= a kernel that accumulates in 16 bits).  Built next to the product library;
# tests select it with DEEPHUMOR_HIP_LIB.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/scratch/degraded
mkdir -p $D/csrc $D/obj
cp $R/deephumor_amd/csrc/* $D/csrc/
python3 - "$D/csrc/lstm_wreg.hip" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
# 16-bit ACCUMULATION: the accumulator is rounded to bf16 after every 32-k MFMA step
old = "        acc[i] = Op16<OT>::mfma(wf[f], fa[t % (PF + 1)], acc[i]);"
new = ("        acc[i] = Op16<OT>::mfma(wf[f], fa[t % (PF + 1)], acc[i]);\n"
       "        for (int e_ = 0; e_ < 4; ++e_) acc[i][e_] = bf16_to_f32(f32_to_bf16(acc[i][e_]));")
assert old in s
open(p, "w").write(s.replace(old, new))
PY
sed -i 's#"../../include/deephumor_hip.h"#"'$R'/include/deephumor_hip.h"#' $D/csrc/common.h
objs=""
for f in $D/csrc/*.hip; do
  o=$D/obj/$(basename ${f%.hip}).o
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-comment -c $f -o $o &
  objs="$objs $o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libdegraded.so $objs
echo built $D/libdegraded.so
