"""The committed evidence bench.py quotes (profiles/instmix.json, profiles/hbm_traffic.json) must exist for the kernels the headline
legs time, carry the fields `roofline.valu` is computed from, and name source summaries that are in the tree -- a round that changes a
kernel has to re-key them.  CPU only: bench.py's helpers are imported, nothing is launched."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_tests", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _sources_exist(text):
    names = re.findall(r"profiles/[A-Za-z0-9_.]+\.(?:txt|json|md)", text)
    assert names, text
    for n in names:
        assert os.path.exists(os.path.join(ROOT, n)), f"{n} is quoted as evidence but is not in the tree"


def test_headline_kernels_have_instruction_mix_and_traffic_entries():
    b = _bench()
    mix = json.load(open(os.path.join(ROOT, "profiles", "instmix.json")))
    traffic = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    for waves, norm in ((5, True), (4, False), (8, True)):
        k = b.kernel_name(waves, "float64", norm, True)
        assert k in mix, k
        ev = mix[k]
        _sources_exist(ev["source"])
        assert 600 < ev["valu_instructions_per_64_drone_step"] < 1400
    # the driver's own launch (K = 20) and the default line's (K = 64) of the headline kernel: PMC entries, not the fitted model
    for K in (20, 64):
        t, src = b.traffic_per_launch("reaching", 32768, "float64", True, K, 5)
        assert src.startswith("pmc:") and t > b.algo_bytes_per_launch(32768, K, True) * 0.9, (K, t, src)
        _sources_exist(traffic[src[4:]]["source"])


def test_valu_bound_is_computed_from_the_committed_counter_pass():
    b = _bench()
    k = b.kernel_name(5, "float64", True, True)
    v20 = b.valu_bound(k, 32768, 20, 35.6 / 20)
    v64 = b.valu_bound(k, 32768, 64, 1.44)
    for v in (v20, v64):
        assert v is not None and v["tiles_per_cu"] == 2.0 and 3.5 < v["valu_cycles_per_inst"] < 5.5
        floor = v["tiles_per_cu"] * v["valu_insts_per_tile_step"] * v["valu_cycles_per_inst"] / 4.0 / (v["shader_clock_ghz"] * 1e3)
        assert abs(floor - v["valu_floor_us_per_step"]) < 1e-3 and 0.3 < v["valu_frac"] < 1.0
    assert "r06_instmix_k20" in v20["counters_from"]              # the driver's launch quotes the pass of the driver's own command
    assert v20["valu_insts_per_tile_step"] != v64["valu_insts_per_tile_step"]
    assert b.valu_bound("no such kernel", 32768, 20, 1.0) is None


def _lookup(path, key):
    full = os.path.join(ROOT, path)
    if key.startswith("regex:"):
        m = re.search(key[len("regex:"):], open(full).read(), flags=re.M)
        assert m, (path, key)
        return float(m.group(1))
    obj = json.load(open(full))
    for part in (key.split("|") if "|" in key else key.split(".")):
        obj = obj[part]
    return float(obj)


def test_prose_quotes_the_committed_numbers():
    """VERDICT r05 next #4a: README / DESIGN section 5 quote numbers; every such number listed in profiles/r06_claims.json must (1) appear in
    the documents as quoted and (2) contain the value of the committed final file it names -- not the best box of an earlier round."""
    claims = json.load(open(os.path.join(ROOT, "profiles", "r06_claims.json")))["claims"]
    assert len(claims) >= 20
    docs = {}
    for c in claims:
        for d in c["docs"]:
            docs.setdefault(d, open(os.path.join(ROOT, d)).read())
            assert c["text"] in docs[d], f"{d} does not quote {c['text']!r}"
        for path, key in c["where"]:
            v = _lookup(path, key)
            assert c["lo"] <= v <= c["hi"], f"{c['text']!r}: {path} {key} = {v} is outside [{c['lo']}, {c['hi']}]"
