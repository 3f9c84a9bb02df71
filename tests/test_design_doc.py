"""DESIGN.md quotes measurements only through blocks generated from the committed evidence ("DESIGN §5 misquotes its own
evidence"): tools/gen_design_tables.py --check regenerates the blocks from profiles/bench_rNN, rNN_kernel_stats_*.csv, rNN_pmc_*.json and
rNN_parity_report.json and fails if the document differs."""
import re
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def test_generated_blocks_match_the_committed_evidence():
    r = subprocess.run([sys.executable, str(REPO / "tools" / "gen_design_tables.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_design_is_the_current_state_document():
    doc = (REPO / "DESIGN.md").read_text()
    for name in ("performance", "parity"):
        m = re.search(rf"<!-- BEGIN GENERATED {name} -->\n(.*?)\n<!-- END GENERATED {name} -->", doc, re.S)
        assert m and len(m.group(1)) > 200, f"generated block {name} is missing or empty"
    assert (REPO / "HISTORY.md").exists()
    # (the two generated blocks are ≈ 17 KB of it since round 4: 33 bench lines × 18 columns and the per-family parity table)
    assert len(doc) < 64_000, "DESIGN.md is the current-state document; narrative belongs in HISTORY.md"
    # the parity paragraph of the document is the report's own summary: its Float32 / Float64 worst well-conditioned errors meet the bounds
    blk = re.search(r"<!-- BEGIN GENERATED parity -->\n(.*?)\n<!-- END GENERATED parity -->", doc, re.S).group(1)
    assert "Rows whose worst well-conditioned point exceeds the tolerance: 0." in blk
