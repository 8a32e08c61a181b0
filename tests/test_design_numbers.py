"""DESIGN.md stays a description of the current tree: at most 400 lines, and every measured figure in a paragraph that cites
`profiles/<file>` is found in one of the cited files (tools/check_design_numbers.py; VERDICT r03 "next" #7 -- an explanation
must not outlive its evidence).  The checker itself is exercised on a small synthetic document."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_design_numbers.py")


def test_design_md_is_short_and_its_figures_are_in_the_files_it_cites():
    lines = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read().count("\n")
    assert lines <= 400, lines
    r = subprocess.run([sys.executable, TOOL], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " 0 not found" in r.stdout
    # the other two documents that quote measurements with their source
    for doc in ("README.md", "INTEGRATION.md"):
        r = subprocess.run([sys.executable, TOOL, os.path.join(ROOT, doc)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, doc + "\n" + r.stdout[-4000:] + r.stderr[-2000:]


def test_checker_finds_a_stale_figure_and_a_missing_file(tmp_path):
    prof = os.path.join(ROOT, "profiles", "r04_launch_contention.log")
    assert os.path.exists(prof)
    good = tmp_path / "good.md"
    good.write_text("# t\n\nOne thread pays 2.76 µs per launch call [`profiles/r04_launch_contention.log`]; the bound `1e-12` and ≈3 µs are not read-outs.\n",
                    encoding="utf-8")
    r = subprocess.run([sys.executable, TOOL, str(good)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    stale = tmp_path / "stale.md"
    stale.write_text("# t\n\nOne thread pays 9.99 µs per launch call [`profiles/r04_launch_contention.log`].\n\n"
                     "| row | value |\n|---|---|\n| x | 1.23e45 |\n\nElsewhere 4.2 ms [`profiles/no_such_file.log`].\n", encoding="utf-8")
    r = subprocess.run([sys.executable, TOOL, str(stale)], capture_output=True, text=True)
    assert r.returncode == 1
    assert "9.99" in r.stdout and "no_such_file" in r.stdout
    # the table row inherits the citation of the paragraph above its table only when that paragraph is directly above it:
    # here the row follows a cited paragraph, so its figure is checked too
    assert "1.23e45" in r.stdout
