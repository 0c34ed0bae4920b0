"""One fixed seed of every randomised sweep of tools/ (``tools/fuzz_all.sh`` runs them at ~20x these trial counts; the summary
lines of a full run are committed under profiles/r4/fuzz_summary.txt) inside ``-m gpu``: a regression in any swept entry point --
generate / forward against the CPU oracle on random model shapes, the samplers, the scorer, the GEMM / convolution / encoder
kernels against fp32 references, the BeamSearchHelper method surface, the pipeline, the kernel variants and option combinations,
the poisoned-allocator check (no kernel reads memory it did not write) -- turns the driver's
run red.  Each tool prints one JSON record per trial and a last summary line; the bar is zero failing trials."""
import importlib
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 11

SWEEPS = [("fuzz_generate", 12, ["--half"]), ("fuzz_generate", 2, ["--long"]), ("fuzz_generate", 8, ["--r4", "--half"]), ("fuzz_generate", 10, ["--split"]), ("fuzz_forward", 12, []), ("fuzz_forward", 2, ["--big"]),
          ("fuzz_encoder", 4, []), ("fuzz_beam_methods", 15, []), ("fuzz_sampler", 25, []), ("fuzz_scoring", 15, []),
          ("fuzz_gemm", 15, []), ("fuzz_f32x", 24, []), ("fuzz_conv", 8, []), ("fuzz_pipeline", 6, []), ("fuzz_variants", 6, []), ("poison_check", 4, [])]


@pytest.mark.parametrize("tool,trials,extra", SWEEPS, ids=[t + "".join(e) for t, _, e in SWEEPS])
def test_one_seed_of_the_sweep(tool, trials, extra, capsys):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, ROOT)
    mod = importlib.import_module(tool)
    rc = mod.main(["--trials", str(trials), "--seed", str(SEED)] + extra)
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    summary = json.loads(lines[-1])
    failed = [l for l in lines[:-1] if '"ok": true' not in l and "fewer positive" not in l]
    assert rc == 0 and summary["trials"] == trials, summary
    assert summary.get("failures", 0) + summary.get("mismatches_or_errors", 0) == 0, (summary, failed[:3])
    # fuzz_generate reclassifies a caption mismatch as "not comparable" when the ORACLE's own decision gap at the first differing draw
    # is below 1e-6 relative (two ulps of fp32): a judgment call, so it is bounded here, not just printed (<= 1 per seed;
    # the full sweeps saw 3 in ~4,900 trials)
    assert summary.get("fp32_near_ties_not_comparable", 0) <= 1, summary
