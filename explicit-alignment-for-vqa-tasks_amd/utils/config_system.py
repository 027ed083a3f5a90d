"""Config surface of the reference: jsonnet files -> attribute dict, with ``--opts a.b.c=value`` overrides.

Mirrors ``src/utils/config_system.py`` (``get_config_from_json`` :25-41, ``parse_optional_args`` :122-159).  The
``_jsonnet`` binding is not installed offline, so the subset of jsonnet the reference's configs use is evaluated
here: ``local x = expr;`` bindings, ``import 'file'``, object / array literals (quoted or bare keys, trailing
commas), single- or double-quoted strings, numbers, ``true/false/null``, ``//``, ``#`` and ``/* */`` comments,
identifier references, ``+`` on numbers / strings / arrays / objects, and ``std.mergePatch(a, b)`` (RFC 7396) -
which is everything ``configs/**/*.jsonnet`` contains.  Anything else raises ``JsonnetError``.
"""
from __future__ import annotations

import ast
import os
import re
from typing import Any, Dict, List, Optional

from .attrdict import AttrDict


class JsonnetError(ValueError):
    pass


_TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|\#[^\n]*|/\*.*?\*/)
  | (?P<num>-?(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)
  | (?P<str>"(?:[^"\\]|\\.)*"|'(?:[^'\\]|\\.)*')
  | (?P<id>[A-Za-z_][A-Za-z_0-9]*)
  | (?P<op>[{}\[\]():,;=+.])
""", re.X | re.S)


def _tokens(text: str, path: str):
    pos, out = 0, []
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            line = text.count("\n", 0, pos) + 1
            raise JsonnetError(f"{path}:{line}: unsupported jsonnet syntax near {text[pos:pos+20]!r}")
        pos = m.end()
        kind = m.lastgroup
        if kind != "ws":
            out.append((kind, m.group(kind)))
    out.append(("eof", ""))
    return out


def merge_patch(target: Any, patch: Any) -> Any:
    """``std.mergePatch`` (RFC 7396)."""
    if not isinstance(patch, dict):
        return patch
    res = dict(target) if isinstance(target, dict) else {}
    for k, v in patch.items():
        if v is None:
            res.pop(k, None)
        else:
            res[k] = merge_patch(res.get(k), v)
    return res


class _Parser:
    def __init__(self, text: str, path: str, lenient: bool = False):
        self.toks = _tokens(text, path)
        self.i = 0
        self.path = path
        self.lenient = lenient
        self.scopes: List[Dict[str, Any]] = [{}]

    def peek(self):
        return self.toks[self.i]

    def take(self, kind=None, val=None):
        k, v = self.toks[self.i]
        if (kind and k != kind) or (val is not None and v != val):
            raise JsonnetError(f"{self.path}: expected {val or kind}, got {v!r}")
        self.i += 1
        return v

    def lookup(self, name: str):
        for sc in reversed(self.scopes):
            if name in sc:
                return sc[name]
        if self.lenient:
            return {}
        raise JsonnetError(f"{self.path}: unknown identifier {name!r}")

    def parse_file(self):
        v = self.expr()
        self.take("eof")
        return v

    def expr(self):
        k, v = self.peek()
        if k == "id" and v == "local":
            self.take()
            name = self.take("id")
            self.take("op", "=")
            self.scopes[-1][name] = self.expr()
            self.take("op", ";")
            return self.expr()
        left = self.term()
        while self.peek() == ("op", "+"):
            self.take()
            right = self.term()
            if isinstance(left, dict) and isinstance(right, dict):
                left = {**left, **right}
            else:
                left = left + right
        return left

    def term(self):
        k, v = self.peek()
        if k == "num":
            self.take()
            f = float(v)
            return int(f) if re.fullmatch(r"-?\d+", v) else f
        if k == "str":
            self.take()
            return ast.literal_eval(v)
        if k == "op" and v == "{":
            return self.obj()
        if k == "op" and v == "[":
            self.take()
            items = []
            while self.peek() != ("op", "]"):
                items.append(self.expr())
                if self.peek() == ("op", ","):
                    self.take()
            self.take("op", "]")
            return items
        if k == "op" and v == "(":
            self.take()
            e = self.expr()
            self.take("op", ")")
            return e
        if k == "id":
            self.take()
            if v in ("true", "false", "null"):
                return {"true": True, "false": False, "null": None}[v]
            if v == "import":
                rel = ast.literal_eval(self.take("str"))
                return evaluate_file(os.path.join(os.path.dirname(self.path), rel), self.lenient)
            if v == "std":
                self.take("op", ".")
                fn = self.take("id")
                self.take("op", "(")
                args = [self.expr()]
                while self.peek() == ("op", ","):
                    self.take()
                    args.append(self.expr())
                self.take("op", ")")
                if fn == "mergePatch" and len(args) == 2:
                    return merge_patch(args[0], args[1])
                raise JsonnetError(f"{self.path}: std.{fn} is not supported")
            val = self.lookup(v)
            while self.peek() == ("op", "."):       # field access on a local object
                self.take()
                val = val[self.take("id")]
            return val
        raise JsonnetError(f"{self.path}: unexpected token {v!r}")

    def obj(self):
        self.take("op", "{")
        out: Dict[str, Any] = {}
        self.scopes.append({})
        while self.peek() != ("op", "}"):
            k, v = self.peek()
            if k == "id" and v == "local":
                self.take()
                name = self.take("id")
                self.take("op", "=")
                self.scopes[-1][name] = self.expr()
            else:
                key = ast.literal_eval(self.take("str")) if k == "str" else self.take("id")
                self.take("op", ":")
                out[key] = self.expr()
            if self.peek() == ("op", ","):
                self.take()
        self.take("op", "}")
        self.scopes.pop()
        return out


def evaluate_file(path: str, lenient: bool = False) -> Any:
    """``lenient``: unknown identifiers evaluate to ``{}`` instead of raising.  The reference's own
    ``configs/vqa2/base_env.jsonnet`` refers to locals it never defines (``VinVL_features``, ``ocr_features``, ... :103-130),
    which real jsonnet rejects as well; lenient mode lets those files be read for the keys that matter."""
    with open(path) as f:
        return _Parser(f.read(), path, lenient).parse_file()


def evaluate_snippet(text: str, path: str = "<snippet>") -> Any:
    return _Parser(text, path).parse_file()


def get_config_from_json(json_file: str, lenient: bool = False):
    """``(config AttrDict, config dict)`` like the reference (:25-41)."""
    d = evaluate_file(json_file, lenient)
    return AttrDict(d), d


def parse_optional_args(config, opts: List[str]):
    """``--opts a.b.c=value`` overrides (:122-159).  Values are parsed as Python literals (the reference ``eval``s them);
    anything that is not a literal stays a string."""
    for opt in opts or []:
        path, value = opt.split("=", 1)
        try:
            value = ast.literal_eval(value)
        except (ValueError, SyntaxError):
            value = str(value)
        keys = path.split(".")
        if len(keys) > 6:
            raise ValueError("Support up to depth=6. Please do not hierarchy the config file too deep.")
        node = config
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = value
    return config


def load_config(path: str, opts: Optional[List[str]] = None, mode: str = "train", experiment_name: str = ""):
    """The part of ``process_config`` (:43-120) the hot path needs: evaluate, apply ``--opts``, derive the paths."""
    config, _ = get_config_from_json(path)
    config.mode = mode
    if experiment_name:
        config.experiment_name = experiment_name
    parse_optional_args(config, opts or [])
    exp = config.get("EXPERIMENT_FOLDER") or os.path.join(os.path.dirname(os.path.abspath(path)), "..", "..", "Experiments")
    name = config.get("experiment_name", "default")
    config.log_path = os.path.join(exp, name, mode)
    config.experiment_path = os.path.join(exp, name)
    config.saved_model_path = os.path.join(exp, name, "train", "saved_model")
    return config
