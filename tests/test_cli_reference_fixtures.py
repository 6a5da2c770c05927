"""N1's config layer against fixtures generated from the REFERENCE's own modules (`tests/golden/make_cli_fixtures.py` loads
`/root/reference/aggfly/cli/config.py` and `cli/preprocess.py` standalone in the build container and writes
`tests/golden/cli_fixtures.json`): for the shapes of both example YAMLs (area and pop — BASELINE configs[0] and [2]) and five
dozen edge configs, `aggfly_amd.cli.config.parse_config` must produce the reference's `RunConfig` fields,
`to_aggregator_dict()`, `resolved_paths()`, `describe()` text and — for invalid configs — the reference's FULL error list;
`aggfly_amd.cli.preprocess.resolve` must accept / refuse exactly the expressions the reference does, with its values and
messages.  Reference: `cli/config.py:98,214-386`, `cli/preprocess.py:143`, `tests/test_cli.py:280-364`."""
import copy
import dataclasses
import json
import os

import numpy as np
import pytest

from aggfly_amd.cli import config as cfgmod, preprocess as ppmod

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cli_fixtures.json")))

# the one deliberate extension of the schema: `aggregate.engine: hip` (and `weights.table`); the reference's message lists
# its own three engines
ENGINE_LIST_REF, ENGINE_LIST_HERE = "['auto', 'dask', 'numba']", "['auto', 'dask', 'hip', 'numba']"


def _jsonable(x):
    if isinstance(x, np.ndarray):
        return {"__ndarray__": x.tolist(), "dtype": str(x.dtype)}
    if isinstance(x, dict):
        return {str(k): _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    if isinstance(x, np.generic):
        return x.item()
    return x


@pytest.mark.parametrize("name", sorted(FIX["config"]))
def test_parse_config_matches_the_reference(name):
    ent = FIX["config"][name]
    raw = copy.deepcopy(ent["raw"])
    if ent["ok"]:
        rc = cfgmod.parse_config(raw)
        got = _jsonable(dataclasses.asdict(rc))
        for k, want in ent["fields"].items():                  # every field the reference's RunConfig has (this one adds weights_table)
            assert got[k] == want, (k, got[k], want)
        assert set(got) - set(ent["fields"]) <= {"weights_table"}
        assert rc.templated == ent["templated"] and rc.resolved_paths() == ent["resolved_paths"]
        assert _jsonable(rc.to_aggregator_dict()) == ent["aggregator"]
        assert cfgmod.describe(rc) == ent["describe"]
        assert cfgmod.check_paths(rc) == ent["check_paths"]
        return
    with pytest.raises(cfgmod.ConfigError) as exc:
        cfgmod.parse_config(raw)
    if ent.get("crash"):
        # the reference itself falls over on these malformed step lists (its multi-dd guard unpacks them after validation):
        # here they are ordinary validation errors — the only requirement is a ConfigError that names the variable
        assert any("aggregate.variables" in e for e in exc.value.errors), exc.value.errors
        return
    want = [e.replace(ENGINE_LIST_REF, ENGINE_LIST_HERE) for e in ent["errors"]]
    assert exc.value.errors == want


@pytest.mark.parametrize("expr", sorted(FIX["preprocess"]["cases"]))
def test_preprocess_expressions_match_the_reference(expr):
    ent = FIX["preprocess"]["cases"][expr]
    x = np.array(FIX["preprocess"]["input"])
    if ent["ok"]:
        with np.errstate(all="ignore"):
            got = np.asarray(ppmod.resolve(expr)(x.copy()), dtype=float)
        assert got.tolist() == ent["value"]
        return
    with pytest.raises(ppmod.PreprocessError) as exc:
        ppmod.resolve(expr)
    assert str(exc.value) == ent["error"]


def test_preprocess_resolve_branches_match_the_reference(tmp_path):
    misc = FIX["preprocess"]["misc"]
    assert (ppmod.resolve(None, None) is None) == misc["none"]
    for label, args in (("both", ("x - 1", "prep.py:f")), ("not_a_string", (3, None)), ("from_no_colon", (None, "prep.py")),
                        ("from_missing_file", (None, "/no/such/file.py:clean"))):
        with pytest.raises(ppmod.PreprocessError) as exc:
            ppmod.resolve(*args)
        assert str(exc.value) == misc[label]["error"], label
