"""The GRC side of the drop-in contract (no GPU): `grc/generate.py` must emit, for every reference block that has a drop-in,
the same block id, parameter ids and `make:` call as the reference's own *.block.yml, and for the `txOFDM_*` blocks -- which
the reference only instantiates in a flowgraph -- every parameter that flowgraph sets.  The expectations are data recorded
from the reference (tests/golden/grc_contract.json, tests/golden/gen_golden_grc.py).  The committed *.block.yml files must be
exactly what the generator emits, and every `make:` must name a class that exists with a matching constructor."""
import importlib
import inspect
import json
import os
import re
import subprocess
import sys

import pytest
import yaml

from conftest import GOLDEN, PKG

GRC_DIR = os.path.join(PKG, "grc")
CONTRACT = json.load(open(os.path.join(GOLDEN, "grc_contract.json")))


@pytest.fixture(scope="module")
def generated(tmp_path_factory):
    out = tmp_path_factory.mktemp("grc")
    r = subprocess.run([sys.executable, os.path.join(GRC_DIR, "generate.py"), "--out", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    docs = {}
    for f in sorted(os.listdir(out)):
        assert f.endswith(".block.yml")
        docs[f] = (open(os.path.join(out, f)).read(), yaml.safe_load(open(os.path.join(out, f))))
    return docs


def test_every_contract_block_is_generated(generated):
    ids = {d["id"] for _, d in generated.values()}
    want = {k for k in CONTRACT if not k.startswith("__")}
    assert want <= ids, want - ids
    for f, (_, d) in generated.items():
        assert f == d["id"] + ".block.yml" and d["file_format"] == 1


@pytest.mark.parametrize("bid", sorted(k for k, v in CONTRACT.items() if "make" in v))
def test_reference_block_descriptions_are_reproduced(generated, bid):
    want = CONTRACT[bid]
    d = generated[bid + ".block.yml"][1]
    assert [p["id"] for p in d["parameters"]] == want["params"]
    assert re.sub(r"\s+", "", d["templates"]["make"]) == want["make"]
    assert len(d.get("inputs") or []) == want["n_inputs"] and len(d.get("outputs") or []) == want["n_outputs"]
    assert d["templates"]["imports"] == "import " + want["make"].split(".", 1)[0]


@pytest.mark.parametrize("bid", sorted(k for k, v in CONTRACT.items() if "params_set_by_flowgraph" in v))
def test_flowgraph_only_blocks_take_the_flowgraphs_parameters(generated, bid):
    d = generated[bid + ".block.yml"][1]
    have = [p["id"] for p in d.get("parameters") or []]
    assert sorted(have) == CONTRACT[bid]["params_set_by_flowgraph"]
    assert d["templates"]["make"].replace("\n", " ").startswith("txOFDM." + bid.split("_", 1)[1] + "(")


def test_flowgraph_connections_have_matching_port_types(generated):
    port = lambda bid, side: (generated[bid + ".block.yml"][1].get(side) or [None])[0]   # noqa: E731
    for src, dst in CONTRACT["__txOFDM_connections__"]:
        assert port(src, "outputs")["dtype"] == port(dst, "inputs")["dtype"], (src, dst)


def test_committed_files_are_the_generators_output(generated):
    committed = sorted(f for f in os.listdir(GRC_DIR) if f.endswith(".block.yml"))
    assert committed == sorted(generated)
    for f in committed:
        assert open(os.path.join(GRC_DIR, f)).read() == generated[f][0], f + " is stale: run grc/generate.py"


def test_make_templates_name_real_constructors(generated):
    for _, d in generated.values():
        make = re.sub(r"\s+", "", d["templates"]["make"])
        m = re.fullmatch(r"(\w+)\.(\w+)\((.*)\)", make)
        assert m, make
        mod = importlib.import_module(m.group(1))
        cls = getattr(mod, m.group(2))
        args = [a for a in m.group(3).split(",") if a]
        assert all(re.fullmatch(r"\$\{\w+\}", a) for a in args)
        sig = inspect.signature(cls.__init__)
        pos = [p for p in list(sig.parameters.values())[1:] if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
        required = [p for p in pos if p.default is p.empty]
        assert len(required) <= len(args) <= len(pos), (make, str(sig))
