#!/usr/bin/env python3
"""Pin the drop-in boundary to the REFERENCE'S OWN call sites (build container only: reads /root/reference).

1. Parses the reference's gaussian_renderer/__init__.py, train.py and render.py with `ast` and extracts, mechanically,
   what they pass to and read from the operator boundary:
     * the names imported from `diff_gaussian_rasterization` (gaussian_renderer/__init__.py:15)
     * the keyword names, in order, of every `GaussianRasterizationSettings(...)`, `GaussianRasterizer(...)`,
       `rasterizer(...)` and `rasterizer.visible_filter(...)` call (:145-171, :208-242), per enclosing function
     * the signatures of render / prefilter_voxel / generate_neural_gaussians (argument names and defaults)
     * the keys of the dicts render() returns (:174-188) and the keys train.py / render.py read from them
       (train.py:155,188; render.py:57), and the positional / keyword shape of their render(...) / prefilter_voxel(...) calls
     * the argument names of GaussianModel.training_statis (scene/gaussian_model.py:761), the consumer of the means2D gradient
   and writes them to tests/golden/callsites.json -- names and literals only, no source text.
   tests/test_host_golden.py::test_boundary_matches_the_reference_call_sites checks this repository's
   GaussianRasterizationSettings._fields, GaussianRasterizer.forward / visible_filter signatures and renderer.render /
   prefilter_voxel against that file.
2. Imports the reference's `gaussian_renderer` module with THIS repository's `diff_gaussian_rasterization` package first on
   sys.path (the other native modules the reference imports are stubbed as in tools/make_golden.py) and asserts that the
   module's GaussianRasterizer / GaussianRasterizationSettings ARE this repository's classes, i.e. that the zero-edit drop-in
   of INTEGRATION.md section 1 resolves at import level.  (Needs the built library: splatco_amd._C loads it on import.)
"""
import ast
import json
import os
import sys
import types

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "callsites.json")


def _literal(node):
    try:
        return ast.literal_eval(node)
    except Exception:
        return ast.unparse(node) if isinstance(node, (ast.Name, ast.Attribute)) else "<expr>"


def _signature(fn):
    a = fn.args
    names = [x.arg for x in a.args]
    defaults = [None] * (len(names) - len(a.defaults)) + [_literal(d) for d in a.defaults]
    return {"args": names, "defaults": {n: d for n, d in zip(names, defaults) if n in names[len(names) - len(a.defaults):]}}


def _calls(tree):
    """[(enclosing function, callee text, positional count (-n - 1 when a *args follows n plain ones), [keyword names])]
    for every call in the module."""
    out = []

    def visit(node, fn):
        for ch in ast.iter_child_nodes(node):
            f2 = ch.name if isinstance(ch, (ast.FunctionDef, ast.AsyncFunctionDef)) else fn
            if isinstance(ch, ast.Call):
                plain = [a for a in ch.args if not isinstance(a, ast.Starred)]
                npos = len(plain) if len(plain) == len(ch.args) else -len(plain) - 1
                out.append((fn, ast.unparse(ch.func), npos, [k.arg for k in ch.keywords if k.arg is not None]))
            visit(ch, f2)
    visit(tree, "<module>")
    return out


def extract():
    res = {"source": "SCUT-BIP-Lab/SplatCo: gaussian_renderer/__init__.py, train.py, render.py, scene/gaussian_model.py (names only)"}
    gr = ast.parse(open(os.path.join(REF, "gaussian_renderer", "__init__.py")).read())
    res["imports_from_diff_gaussian_rasterization"] = sorted(
        a.name for n in ast.walk(gr) if isinstance(n, ast.ImportFrom) and n.module == "diff_gaussian_rasterization" for a in n.names)
    fns = {n.name: n for n in gr.body if isinstance(n, ast.FunctionDef)}
    res["signatures"] = {k: _signature(fns[k]) for k in ("render", "prefilter_voxel", "generate_neural_gaussians")}
    calls = _calls(gr)
    pick = lambda callee: {fn: kw for fn, c, npos, kw in calls if c == callee and not npos}
    res["settings_kwargs"] = pick("GaussianRasterizationSettings")          # per enclosing function
    res["rasterizer_ctor_kwargs"] = pick("GaussianRasterizer")
    res["rasterizer_call_kwargs"] = pick("rasterizer")
    res["visible_filter_kwargs"] = pick("rasterizer.visible_filter")
    assert set(res["settings_kwargs"]) == {"render", "prefilter_voxel"} and set(res["rasterizer_call_kwargs"]) == {"render"}
    assert set(res["visible_filter_kwargs"]) == {"prefilter_voxel"}
    # literals the call sites fix (sh_degree = 1, prefiltered = False; shs = None, cov3D_precomp = None in render)
    lits = {}
    for n in ast.walk(gr):
        if isinstance(n, ast.Call) and ast.unparse(n.func) in ("GaussianRasterizationSettings", "rasterizer"):
            for k in n.keywords:
                if isinstance(k.value, ast.Constant):
                    lits.setdefault(ast.unparse(n.func), {})[k.arg] = k.value.value
    res["literal_kwargs"] = lits
    # result dicts of render(): one per return statement
    res["render_result_keys"] = [[_literal(k) for k in r.value.keys] for r in ast.walk(fns["render"])
                                 if isinstance(r, ast.Return) and isinstance(r.value, ast.Dict)]
    # consumers
    consumers = {}
    for rel in ("train.py", "render.py"):
        tree = ast.parse(open(os.path.join(REF, rel)).read())
        keys = sorted({n.slice.value for n in ast.walk(tree) if isinstance(n, ast.Subscript) and isinstance(n.slice, ast.Constant)
                       and isinstance(n.slice.value, str) and
                       ((isinstance(n.value, ast.Name) and n.value.id == "render_pkg") or
                        (isinstance(n.value, ast.Call) and ast.unparse(n.value.func) == "render"))})
        cs = [{"in": fn, "callee": c, "positional": npos if npos >= 0 else -npos - 1, "star_args": npos < 0, "keywords": kw}
              for fn, c, npos, kw in _calls(tree) if c in ("render", "prefilter_voxel")]
        consumers[rel] = {"result_keys_read": keys, "calls": cs}
    res["consumers"] = consumers
    gm = ast.parse(open(os.path.join(REF, "scene", "gaussian_model.py")).read())
    ts = next(n for n in ast.walk(gm) if isinstance(n, ast.FunctionDef) and n.name == "training_statis")
    res["training_statis_args"] = [a.arg for a in ts.args.args]
    tr = ast.parse(open(os.path.join(REF, "train.py")).read())
    res["training_statis_call_positional"] = [len(n.args) for n in ast.walk(tr) if isinstance(n, ast.Call)
                                               and ast.unparse(n.func).endswith(".training_statis")]
    return res


def import_reference_renderer_against_this_repo():
    """The reference's gaussian_renderer imported with this repository's diff_gaussian_rasterization on the path."""
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)                      # diff_gaussian_rasterization/ of THIS repository wins

    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m

    class _Shaped:
        def __class_getitem__(cls, item):
            return cls

    class _Dummy:
        def __init__(self, *a, **k):
            pass
    stub("torch_scatter", scatter_max=None)
    stub("simple_knn")
    stub("simple_knn._C", distCUDA2=None)
    stub("plyfile", PlyData=_Dummy, PlyElement=_Dummy)
    stub("cv2")
    stub("_gridcreater")
    stub("_gridencoder")
    stub("kornia", create_meshgrid=None)
    stub("jaxtyping", Shaped=_Shaped)
    assert "diff_gaussian_rasterization" not in sys.modules
    import gaussian_renderer as ref_gr              # the reference's module (from /root/reference)
    import diff_gaussian_rasterization as ours
    import splatco_amd.rasterizer as R
    assert os.path.realpath(ref_gr.__file__).startswith(REF), ref_gr.__file__
    assert os.path.realpath(ours.__file__).startswith(ROOT), ours.__file__
    assert ref_gr.GaussianRasterizer is R.GaussianRasterizer and ref_gr.GaussianRasterizationSettings is R.GaussianRasterizationSettings
    # the keyword call the reference makes constructs our settings record (no device needed for the NamedTuple)
    import inspect
    kw = extract()["settings_kwargs"]["render"]
    rs = ref_gr.GaussianRasterizationSettings(**{k: 0 for k in kw})
    assert list(rs._fields) == kw
    sig = inspect.signature(ref_gr.GaussianRasterizer.forward)
    assert set(extract()["rasterizer_call_kwargs"]["render"]) <= set(sig.parameters)
    return True


def main():
    res = extract()
    res["import_resolves_to_this_repo"] = import_reference_renderer_against_this_repo()
    with open(OUT, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print(json.dumps(res, indent=1, sort_keys=True))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
