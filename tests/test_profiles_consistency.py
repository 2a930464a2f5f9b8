"""The kept evidence under profiles/ agrees with the sources at HEAD.

`bench.py` prices a kernel with the PMC counts of `profiles/pmc.json` only when the entry's `src_hash` equals the
hash of the kernel's sources now (`tools/kernel_sources.py`); otherwise the line says `"stale": true` and falls back to
the HBM figure.  Round 4 tracked bench lines produced BEFORE pmc.json was regenerated, so the kept headline line
disagreed with the design table.  Guards: the hash ignores comments / whitespace; every pmc.json entry is current at
HEAD (and profiles/README.md lists exactly those keys); no tracked bench line outside archive/ is stale."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_sources as ks  # noqa: E402


def test_source_hash_ignores_comments_and_whitespace():
    code = 'int a = 1;  // one\n/* block\n comment */\nconst char* s = "// not a comment";\n\n\tint b = a  +  2;\n'
    same = 'int a = 1;\nconst char* s = "// not a comment"; int b = a + 2; // tail\n'
    other = 'int a = 1;\nconst char* s = "// not a comment"; int b = a + 3;\n'
    assert ks.strip_comments(code) == ks.strip_comments(same)
    assert ks.strip_comments(code) != ks.strip_comments(other)
    assert '"// not a comment"' in ks.strip_comments(code)


def test_every_pmc_entry_is_current_at_head():
    with open(os.path.join(ROOT, "profiles", "pmc.json")) as f:
        pmc = json.load(f)
    assert pmc
    stale = [k for k, v in pmc.items() if v.get("src_hash") != ks.source_hash(k)]
    assert not stale, f"profiles/pmc.json entries profiled on other sources than HEAD's: {stale} (re-run " \
                      f"tools/profile_bench.sh + tools/make_pmc_json.py for them)"
    for k, v in pmc.items():
        assert k.endswith(f"@{v['num_envs']}"), k
        src = os.path.join(ROOT, v["source"])
        assert os.path.exists(src), f"{k}: {v['source']} (the rocprofv3 summary the counts come from) is not tracked"
    readme = open(os.path.join(ROOT, "profiles", "README.md")).read()
    named = set(re.findall(r"`([A-Za-z0-9]+StepKernel[^`]*@\d+)`", readme))
    assert named == set(pmc), (sorted(named - set(pmc)), sorted(set(pmc) - named))


def test_no_tracked_bench_line_is_stale():
    bad = []
    for path in glob.glob(os.path.join(ROOT, "profiles", "*.json")) + glob.glob(os.path.join(ROOT, "profiles", "*.jsonl")):
        for line in open(path):
            if '"stale": true' in line:
                bad.append(os.path.basename(path))
                break
    assert not bad, f"bench lines written before pmc.json was regenerated (move to archive/ or regenerate): {bad}"
