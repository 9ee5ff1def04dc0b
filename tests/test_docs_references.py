"""The documents cite measurement files under profiles/ and sources under tools/, tests/, restir_amd/: every cited path must exist
(README.md, INTEGRATION.md and the round-5 part of EXPERIMENTS.md; brace lists like r05_config{3,4,5}_x.csv are expanded)."""
import itertools
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _expand(path):
    parts = re.split(r"\{([^{}]*)\}", path)
    choices = [p.split(",") if i % 2 else [p] for i, p in enumerate(parts)]
    return ["".join(c) for c in itertools.product(*choices)]


def _cited(text):
    out = set()
    for m in re.finditer(r"`((?:profiles|tools|tests|restir_amd|oracle|include)/[A-Za-z0-9_./{},*-]+)`", text):
        p = m.group(1).rstrip(".,")
        if "*" in p or p.endswith("/"):
            continue
        out.update(_expand(p))
    # profile files cited by their bare name (`r05_config3_hbm_counters.json`, `_kernel_stats.csv` continuations are not expanded)
    for m in re.finditer(r"`(r0[1-9]_[A-Za-z0-9_{},-]+\.(?:log|json|csv|txt))`", text):
        out.update("profiles/" + q for q in _expand(m.group(1)))
    return out


def test_cited_files_exist():
    texts = []
    for name in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        with open(os.path.join(ROOT, name)) as fh:
            texts.append(fh.read())
    with open(os.path.join(ROOT, "EXPERIMENTS.md")) as fh:
        e = fh.read()
    texts.append(e[e.index("## Round 5"):e.index("## Round 4")])
    missing = []
    for t in texts:
        for p in sorted(_cited(t)):
            full = os.path.join(ROOT, p)
            # built artefacts and generated files are cited too (librestir_hip.so, oracle/_ref/...): only sources and profiles are checked
            if p.endswith(".so") or "/_ref/" in p or p.startswith("restir_amd/host/") and "." not in os.path.basename(p):
                continue
            if not os.path.exists(full):
                missing.append(p)
    assert not missing, missing
