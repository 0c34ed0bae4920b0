#!/usr/bin/env bash
# round 6, call 9: the round's profile collection (tools/collect_profiles.sh: rocprofv3 kernel traces, PMC passes, bench lines, probes)
cd $GRAFT_REPO_ROOT
ROUND=r6 bash tools/collect_profiles.sh > gpurun_out/r6_collect.log 2>&1
tail -5 gpurun_out/r6_collect.log
python3 -c "
import json
d = json.load(open('gpurun_out/profiles_r6/bench_default_bf16.json'))
print({k: d[k] for k in ('value', 'ms_per_step', 'value_pipelined', 'whole_step_mfma_frac')}, d['c3']['ms_per_step'])
"
