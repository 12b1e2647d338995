"""CPU: the oracle reproduces the committed digests of its own outputs (tests/golden/oracle_digests.json, written by
tools/gen_oracle_digests.py).  This is a drift check of the restatement and of the synthetic input generator across
rounds and machines -- not a pin to the reference, which ships no fixtures (parity unpinned)."""
import importlib.util
import json
import os
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_outputs_have_not_drifted():
    spec = importlib.util.spec_from_file_location("gen_oracle_digests", os.path.join(ROOT, "tools", "gen_oracle_digests.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_digests.json")))
    got = json.loads(json.dumps(mod.compute()))
    assert sorted(got) == sorted(want)
    # the generator goes through numpy's libm-backed float functions: should another machine produce different input bytes,
    # the output digests say nothing about the oracle
    if any(got[k] != want[k] for k in want if k.startswith("input_")):
        pytest.skip("synthetic inputs differ on this machine (numpy / libm), digests not comparable")
    for k in want:
        assert got[k] == want[k], k
