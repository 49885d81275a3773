#!/usr/bin/env python3
"""Surface fixture: every attribute of the modules `tools`, `models`, `networks` -- and of the objects built from
them (RSSM via `.dynamics`, WorldModel via `_wm` / `_world_model` / `world_model`, ImagBehavior via
`_task_behavior` / `_behavior`) -- that the reference's drivers and add-ons touch.  Build container only: reads the
reference's SOURCE TEXT with `ast` (nothing is imported or copied) and writes names + use sites to
tests/golden/surface.json.

    python tests/golden/make_surface.py
"""
import ast
import json
import os

REF = "/root/reference"
FILES = ["dreamer.py", "exploration.py", "scm_world_model.py", "causal_VAE.py", "main_with_causal.py"]
MODULES = ("tools", "models", "networks", "expl")  # dreamer.py:16 `import exploration as expl`
OBJECTS = {"dynamics": "RSSM", "rssm": "RSSM", "_rssm": "RSSM", "_wm": "WorldModel", "_world_model": "WorldModel", "world_model": "WorldModel",
           "_task_behavior": "ImagBehavior", "_behavior": "ImagBehavior"}


def tail_name(node):
    """x.y.z -> 'z' for Attribute chains, 'x' for a Name."""
    if isinstance(node, ast.Attribute):
        return node.attr
    if isinstance(node, ast.Name):
        return node.id
    return None


def main():
    out = {m: {} for m in MODULES}
    out.update({c: {} for c in set(OBJECTS.values())})
    for fn in FILES:
        path = os.path.join(REF, fn)
        tree = ast.parse(open(path).read(), filename=fn)
        for node in ast.walk(tree):
            if not isinstance(node, ast.Attribute):
                continue
            base = node.value
            if isinstance(base, ast.Name) and base.id in MODULES:
                out[base.id].setdefault(node.attr, []).append(f"{fn}:{node.lineno}")
            else:
                owner = tail_name(base)
                if owner in OBJECTS:
                    out[OBJECTS[owner]].setdefault(node.attr, []).append(f"{fn}:{node.lineno}")
    for k in out:
        out[k] = {a: sorted(set(v)) for a, v in sorted(out[k].items())}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "surface.json")
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print({k: len(v) for k, v in out.items()}, "->", dst)


if __name__ == "__main__":
    main()
