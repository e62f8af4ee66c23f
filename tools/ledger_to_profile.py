#!/usr/bin/env python3
"""gpurun_out/<ledger>.jsonl (written by tests/conftest.py::bounded under GYMRL_TOL_LEDGER) -> profiles/rNN_trace_tolerances.json:
the observed drift of every multi-step trace bound of an MI355X run next to the bound the test enforces.
usage: python tools/ledger_to_profile.py <ledger.jsonl> <out.json>"""
import json
import subprocess
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
best = {}
for r in rows:
    k = r["name"]
    if k not in best or r["observed"] > best[k]["observed"]:
        best[k] = r
try:
    head = subprocess.check_output(["git", "rev-parse", "HEAD"], text=True).strip()
except Exception:
    head = None
out = {"what": "max observed error per trace bound (|got - want| / max(1, |want|) unless the test says otherwise) on one MI355X run of "
               "pytest -m gpu with GYMRL_TOL_LEDGER set; fixtures are the reference's own train() / update() runs (tests/golden/make_golden.py)",
       "git_head": head, "bounds": sorted(best.values(), key=lambda r: r["name"])}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(f"{len(best)} bounds; worst observed / bound = {max(r['observed'] / r['bound'] for r in best.values()):.3f}")
