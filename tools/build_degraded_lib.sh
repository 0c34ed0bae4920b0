#!/usr/bin/env bash
# A deliberately degraded library for the "the gates turn red" check: the register-stationary LSTM step rounds its
# gate pre-activations to bf16 before the cell update (= a kernel that accumulated in 16 bits).  Built next to the product library;
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
old = "const float gi = acc[i][0] + b4.x, gf = acc[i][1] + b4.y, gg = acc[i][2] + b4.z, go = acc[i][3] + b4.w;"
new = ("const float gi = bf16_to_f32(f32_to_bf16(acc[i][0] + b4.x)), gf = bf16_to_f32(f32_to_bf16(acc[i][1] + b4.y)), "
       "gg = bf16_to_f32(f32_to_bf16(acc[i][2] + b4.z)), go = bf16_to_f32(f32_to_bf16(acc[i][3] + b4.w));")
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
