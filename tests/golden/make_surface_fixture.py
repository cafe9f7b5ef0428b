"""Generates tests/golden/surface.json: the Python surface of the reference's hot path, read with `ast` only.

    python tests/golden/make_surface_fixture.py          (build container only: /root/reference does not travel)

Nothing of the reference is imported or executed (TensorFlow is not installed here): the files are parsed as text.  The JSON
holds, per class of the path ("positional order and defaults must match"):
  * `methods`: name -> list of [argument name, default as source text or null] in positional order (+ *args / **kwargs markers);
  * `properties`: names declared with @property;
  * `kwargs`: the hyper-parameter keys the class reads with `kwargs.get(key[, default])` / `kwargs[key]` -> default as source text.
It is data about the interface (names, order, literal defaults) — no function bodies, no source lines.
tests/test_surface.py holds the host mirror (myconvnet_amd/*.py) to it.
"""
import ast
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
# file -> classes of the path whose surface the mirror keeps
FILES = {
    'convnet.py': None,                         # None = every class of the file
    'optimizers.py': ['Optimizer', 'MomentumOptimizer'],
    'models/resnet_v1_5.py': None,
    'models/vggnet.py': None,
    'models/efficientnet.py': None,
    'models/resnet_v1_5_dilated.py': None,
    'models/deeplabv3plus.py': None,
    'segmentation/segnet.py': None,
    'evaluators.py': ['Evaluator', 'AccuracyEvaluator', 'AccuracyTopNEvaluator', 'AccuracyTop1Evaluator', 'AccuracyTop5Evaluator', 'ErrorEvaluator', 'MeanIoUEvaluator'],
}


def src(node):
    return None if node is None else ast.unparse(node)


def signature(fn):
    a = fn.args
    pos = list(a.posonlyargs) + list(a.args)
    defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
    out = [[p.arg, src(d)] for p, d in zip(pos, defaults)]
    if a.vararg is not None:
        out.append(['*' + a.vararg.arg, None])
    for p, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append([p.arg, src(d)])
    if a.kwarg is not None:
        out.append(['**' + a.kwarg.arg, None])
    return out


def kwargs_reads(cls):
    """keys read from a `kwargs` dict (or self._parameters / self._curr_parameters style aliases are NOT followed: the reference
    reads its hyper-parameters with kwargs.get in the methods that receive **kwargs)"""
    found = {}
    for node in ast.walk(cls):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == 'get' \
                and isinstance(node.func.value, ast.Name) and node.func.value.id == 'kwargs' and node.args \
                and isinstance(node.args[0], ast.Constant) and isinstance(node.args[0].value, str):
            key = node.args[0].value
            d = src(node.args[1]) if len(node.args) > 1 else 'None'
            found.setdefault(key, [])
            if d not in found[key]:
                found[key].append(d)
        elif isinstance(node, ast.Subscript) and isinstance(node.value, ast.Name) and node.value.id == 'kwargs' \
                and isinstance(node.slice, ast.Constant) and isinstance(node.slice.value, str):
            found.setdefault(node.slice.value, [])
            if '<required>' not in found[node.slice.value]:
                found[node.slice.value].append('<required>')
    return found


def main():
    out = {}
    for rel, classes in FILES.items():
        path = os.path.join(REF, rel)
        if not os.path.exists(path):
            continue
        tree = ast.parse(open(path).read())
        for node in tree.body:
            if isinstance(node, ast.ClassDef) and (classes is None or node.name in classes):
                methods, props = {}, []
                for item in node.body:
                    if isinstance(item, (ast.FunctionDef, ast.AsyncFunctionDef)):
                        decos = [src(d) for d in item.decorator_list]
                        if 'property' in decos:
                            props.append(item.name)
                        elif not any(d.endswith('.setter') for d in decos):
                            methods[item.name] = signature(item)
                out['{}::{}'.format(rel, node.name)] = {
                    'bases': [src(b) for b in node.bases],
                    'methods': methods,
                    'properties': sorted(props),
                    'kwargs': kwargs_reads(node),
                }
    with open(os.path.join(HERE, 'surface.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('surface.json: {} classes, {} methods, {} kwargs keys'.format(
        len(out), sum(len(c['methods']) for c in out.values()), sum(len(c['kwargs']) for c in out.values())))


if __name__ == '__main__':
    main()
