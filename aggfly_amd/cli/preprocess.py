"""``preprocess`` resolution (`aggfly/cli/preprocess.py:1-190` behaviour): a named builtin, a
safe arithmetic expression in the single variable ``x`` (parsed with ``ast`` against an
allowlist — no calls, attributes, subscripts or other names), or a trusted
``path/to/file.py:function`` escape hatch."""
from __future__ import annotations

import ast
import importlib.util
import operator
import os

BUILTINS = {
    "identity": lambda x: x,
    "kelvin_to_celsius": lambda x: x - 273.15,
    "celsius_to_kelvin": lambda x: x + 273.15,
    "pa_to_kpa": lambda x: x / 1000.0,
    "m_to_mm": lambda x: x * 1000.0,
}
_BIN = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv,
        ast.Pow: operator.pow, ast.Mod: operator.mod, ast.FloorDiv: operator.floordiv}
_UN = {ast.UAdd: operator.pos, ast.USub: operator.neg}


class PreprocessError(Exception):
    pass


def _check(node):
    if isinstance(node, ast.Expression):
        return _check(node.body)
    if isinstance(node, ast.BinOp):
        if type(node.op) not in _BIN:
            raise PreprocessError(f"operator {type(node.op).__name__} is not allowed")
        _check(node.left)
        return _check(node.right)
    if isinstance(node, ast.UnaryOp):
        if type(node.op) not in _UN:
            raise PreprocessError(f"unary {type(node.op).__name__} is not allowed")
        return _check(node.operand)
    if isinstance(node, ast.Constant):
        if isinstance(node.value, bool) or not isinstance(node.value, (int, float)):
            raise PreprocessError(f"only numeric constants are allowed, got {node.value!r}")
        return None
    if isinstance(node, ast.Name):
        if node.id != "x":
            raise PreprocessError(f"only the variable 'x' is allowed, got {node.id!r}")
        return None
    raise PreprocessError(f"expression element {type(node).__name__} is not allowed (only arithmetic on 'x' and numbers)")


def _eval(node, x):
    if isinstance(node, ast.Expression):
        return _eval(node.body, x)
    if isinstance(node, ast.BinOp):
        return _BIN[type(node.op)](_eval(node.left, x), _eval(node.right, x))
    if isinstance(node, ast.UnaryOp):
        return _UN[type(node.op)](_eval(node.operand, x))
    if isinstance(node, ast.Constant):
        return node.value
    return x


def compile_expression(expr: str):
    try:
        tree = ast.parse(expr, mode="eval")
    except SyntaxError as e:
        raise PreprocessError(f"could not parse expression {expr!r}: {e.msg}")
    _check(tree)
    if not any(isinstance(n, ast.Name) and n.id == "x" for n in ast.walk(tree)):
        raise PreprocessError(f"expression {expr!r} must use the variable 'x' (e.g. 'x - 273.15')")
    return lambda x: _eval(tree, x)


def load_from_file(spec: str):
    path, _, func = spec.rpartition(":")
    if not path or not func:
        raise PreprocessError(f"preprocess_from must be 'path/to/file.py:function', got {spec!r}")
    if not os.path.exists(path):
        raise PreprocessError(f"preprocess_from file not found: {path}")
    mod_spec = importlib.util.spec_from_file_location("_aggfly_user_preprocess", path)
    mod = importlib.util.module_from_spec(mod_spec)
    mod_spec.loader.exec_module(mod)
    if not hasattr(mod, func) or not callable(getattr(mod, func)):
        raise PreprocessError(f"{path} has no callable {func!r}")
    return getattr(mod, func)


def resolve(preprocess=None, preprocess_from=None):
    if preprocess is not None and preprocess_from is not None:
        raise PreprocessError("set at most one of 'preprocess' and 'preprocess_from'")
    if preprocess_from is not None:
        return load_from_file(preprocess_from)
    if preprocess is None:
        return None
    if not isinstance(preprocess, str):
        raise PreprocessError(f"preprocess must be a builtin name or an expression string, got {type(preprocess).__name__}")
    if preprocess in BUILTINS:
        return BUILTINS[preprocess]
    try:
        return compile_expression(preprocess)
    except PreprocessError:
        if preprocess.isidentifier():            # a bare word: the user meant a builtin
            raise PreprocessError(f"unknown preprocess {preprocess!r}: not a builtin ({', '.join(sorted(BUILTINS))}) "
                                  "and not a valid expression") from None
        raise


def resolve_from_config(config):
    return resolve(config.preprocess, config.preprocess_from)
