"""Build-container check (skipped where /root/reference is absent, e.g. on the GPU box): the reference's own causal
world models -- scm_world_model.WorldModelWithSCM (scm_world_model.py:407) and causal_VAE.CausalVAE_WorldModel
(causal_VAE.py:858) -- import and CONSTRUCT against this repo's `networks` / `tools`.  They stay
the integrator's files; what is pinned here is that the class surface they build on (RSSM / MultiEncoder /
MultiDecoder / MLP constructors and private members, tools.Optimizer) is the reference's.  The reference files are
imported from where they lie, in a child process (its sys.path / sys.modules stay its own); nothing is copied."""
import os
import subprocess
import sys
import textwrap

import pytest

REFERENCE = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, "scm_world_model.py")),
                    reason="the reference checkout is only present in the build container")
def test_reference_causal_world_models_construct_on_these_modules():
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {os.path.join(REPO, "dreamerv3-torch_amd")!r}); sys.path.insert(0, {REPO!r})
        sys.path.append({REFERENCE!r})  # behind the package: networks / tools / models resolve to the MI355X modules
        import torch
        import networks, tools
        from dv3hip import shapes
        assert networks.__file__.startswith({REPO!r}) and tools.__file__.startswith({REPO!r})
        import scm_world_model, causal_VAE
        assert scm_world_model.__file__.startswith({REFERENCE!r}) and causal_VAE.__file__.startswith({REFERENCE!r})
        cfg, obs = shapes.make_config("tiny", "cpu"), shapes.obs_space("tiny")
        a = scm_world_model.WorldModelWithSCM(obs, None, 0, cfg)
        b = causal_VAE.CausalVAE_WorldModel(obs, None, 0, cfg)
        assert isinstance(b.dynamics, networks.RSSM) and isinstance(a.encoder, networks.MultiEncoder)
        assert isinstance(a.dynamics._rssm, networks.RSSM) and isinstance(a.dynamics._rssm._cell, networks.GRUCell)
        for m in (a, b):
            assert isinstance(m._model_opt, tools.Optimizer) and sum(p.numel() for p in m.parameters()) > 0
        print("constructed", type(a).__name__, type(b).__name__)
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "constructed WorldModelWithSCM CausalVAE_WorldModel" in r.stdout
