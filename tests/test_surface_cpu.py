"""Drop-in surface: every attribute of `tools`, `models`, `networks` -- and of the RSSM / WorldModel /
ImagBehavior objects -- that the reference's dreamer.py, exploration.py, scm_world_model.py, causal_VAE.py and
main_with_causal.py touch (tests/golden/surface.json, written by tests/golden/make_surface.py from the reference's
source text in the build container) resolves on the MI355X modules, is on the declared pass-through list (host-side
helpers that stay the integrator's reference code), or is declared outside the accelerated path with a reason."""
import json
import os
import sys
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SURFACE = json.load(open(os.path.join(HERE, "golden", "surface.json")))
# methods of the fork's SCM / counterfactual add-ons: defined by ITS subclasses (scm_world_model.SCMRSSM,
# WorldModelWithSCM) or behind the unreachable `_best_candidate` branch (gotcha 2)
OUTSIDE = {
    "RSSM": {"intervene": "scm_world_model.SCMRSSM method", "remove_intervention": "scm_world_model.SCMRSSM method"},
    "WorldModel": {"intervene": "WorldModelWithSCM method", "remove_intervention": "WorldModelWithSCM method",
                   "counterfactual_imagine": "WorldModelWithSCM method"},
    "ImagBehavior": {"select_counterfactual_actions": "dreamer.py:165 branch on the never-set _best_candidate"},
}


def test_module_attributes_resolve_or_pass_through():
    import exploration
    import models
    import networks
    import tools

    mods = {"tools": tools, "models": models, "networks": networks, "expl": exploration}
    for mname, attrs in SURFACE.items():
        if mname not in mods:
            continue
        for a, sites in attrs.items():
            local = a in vars(mods[mname])
            assert local or (mname == "tools" and a in tools.PASSTHROUGH), f"{mname}.{a} (used at {sites[:3]}) is missing"


def test_object_members_exist():
    import models
    import networks

    classes = {"RSSM": networks.RSSM, "WorldModel": models.WorldModel, "ImagBehavior": models.ImagBehavior}
    instance_attrs = {"WorldModel": {"dynamics", "encoder", "heads", "embed_size"}, "ImagBehavior": {"actor"}}
    # the private members scm_world_model.py:129-165 / causal_VAE.py:1011 reach into: checked on a constructed RSSM
    # (construction is host-side; the modules they name are callable pieces, see tests/test_autograd_gpu.py)
    rssm = networks.RSSM(stoch=4, deter=16, hidden=16, discrete=4, num_actions=3, embed=32, device="cpu")
    instance_attrs["RSSM"] = {a for a in SURFACE["RSSM"] if hasattr(rssm, a)}
    for a in ("_img_in_layers", "_img_out_layers", "_obs_out_layers", "_cell"):
        assert callable(getattr(rssm, a)), a
    for cname, attrs in SURFACE.items():
        if cname not in classes:
            continue
        for a, sites in attrs.items():
            if a in OUTSIDE.get(cname, {}):
                continue
            ok = hasattr(classes[cname], a) or a in instance_attrs.get(cname, set())
            assert ok, f"{cname}.{a} (used at {sites[:3]}) is missing"


def test_pass_through_resolves_from_the_integrators_reference_tools(monkeypatch):
    """Names this module does not define come from the reference tools.py at DV3_REFERENCE_TOOLS (a stand-in file
    here: the real reference never travels with the repo)."""
    import tools

    d = tempfile.mkdtemp()
    with open(os.path.join(d, "tools.py"), "w") as f:
        f.write("class Logger:\n    marker = 'reference'\n\ndef simulate(*a, **k):\n    return 'simulated'\n")
    monkeypatch.setenv("DV3_REFERENCE_TOOLS", d)
    monkeypatch.setattr(tools, "_REFERENCE_TOOLS", None)
    sys.modules.pop("_dv3_reference_tools", None)
    assert tools.Logger.marker == "reference" and tools.simulate() == "simulated"
    assert tools.symlog.__module__ == "tools"  # locally defined names are never shadowed
    with pytest.raises(AttributeError):
        tools.definitely_not_a_reference_name
    monkeypatch.setattr(tools, "_REFERENCE_TOOLS", None)
    monkeypatch.setenv("DV3_REFERENCE_TOOLS", os.path.join(d, "missing"))
    with pytest.raises(AttributeError, match="host-side helper"):
        tools.Logger


def test_optimizer_call_signature_is_the_references():
    import inspect

    import tools

    sig = inspect.signature(tools.Optimizer.__call__)
    assert list(sig.parameters)[:3] == ["self", "loss", "params"]
    assert sig.parameters["retain_graph"].default is True  # tools.py:760
