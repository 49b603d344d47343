"""Fingerprints of the two forwards sample.py rebinds on the UNet (sample.py:33-79 `customforward`, :82-136 `_customforward`).

sgm/modules/attention.py serves that rebinding natively (the fused sampling mode) -- but only for functions that ARE sample.py's: a
function that merely carries the name (a user's edited copy: another blend, an extra hook, attention-map capture) must run its own body.
So the assignment is checked against a fingerprint of the function's SOURCE, taken on the user's machine at rebinding time:

    fingerprint(fn) = sha256 of a canonical walk of the function's AST (node class names, identifiers, constants; no line numbers,
                      no comments, no formatting)

and compared with the values below, which tests/golden/make_golden.py computes from /root/reference/sample.py with the same walk (a hash is
data about the reference, not its text; tests/golden/sample_py_fingerprints.json is the recorded copy, tests/test_host_cpu.py holds the two
equal and -- where the reference tree is present -- recomputes them from sample.py).  A function with no retrievable source, or whose
fingerprint is not listed, is NOT recognised: it is installed like any other instance-level forward and takes the strict module route,
with one warning.  `trust(kind, fn)` lets a caller declare one more function as sample.py's sampling mode (a vendored copy with cosmetic
edits; the stand-ins of tests/golden/sample_py_stub.py)."""
from __future__ import annotations

import ast
import hashlib
import inspect
import textwrap
import warnings
from typing import Optional

# kind -> fingerprints accepted; the first entry of each is /root/reference/sample.py's (tests/golden/sample_py_fingerprints.json)
KNOWN = {
    "block": {"2211bd009ed18fe90cf740951122c28dc953bbc4905ea9d875bf4206c7d72902"},
    "st": {"3049de6004e9243435b0b00ba9b9b146f6e91cca5b8b03376d6bf7380cb8f27e"},
}
_NAMES = {"_customforward": "block", "customforward": "st"}
_warned = set()
_fp_cache = {}  # code object -> fingerprint (sample.py rebinds the same two functions on every block of the UNet)


def _canon(node, out: list) -> None:
    """Canonical pre-order walk: class name, then every populated field in the node's declared order.  Empty lists and None are skipped,
    so fields later Python versions add with empty defaults (`type_params`, `type_comment`) do not move the hash."""
    out.append(type(node).__name__)
    for name in node._fields:
        v = getattr(node, name, None)
        if v is None or (isinstance(v, list) and not v):
            continue
        out.append(name)
        if isinstance(v, list):
            out.append("[")
            for item in v:
                if isinstance(item, ast.AST):
                    _canon(item, out)
                else:
                    out.append(repr(item))
            out.append("]")
        elif isinstance(v, ast.AST):
            _canon(v, out)
        else:
            out.append(repr(v))


def fingerprint_node(node: ast.AST) -> str:
    out: list = []
    _canon(node, out)
    return hashlib.sha256("\x1f".join(out).encode()).hexdigest()


def fingerprint(fn) -> Optional[str]:
    """Fingerprint of a plain function from its source file; None when the source cannot be read or parsed."""
    try:
        tree = ast.parse(textwrap.dedent(inspect.getsource(fn)))
    except (OSError, TypeError, SyntaxError, IndentationError):
        return None
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == getattr(fn, "__name__", None):
            return fingerprint_node(node)
    return None


def trust(kind: str, fn) -> str:
    """Declare `fn` to be sample.py's `_customforward` (kind "block") or `customforward` (kind "st"): its fingerprint joins KNOWN."""
    fp = fingerprint(getattr(fn, "__func__", fn))
    if fp is None:
        raise ValueError(f"{fn!r}: no source to fingerprint")
    KNOWN[kind].add(fp)
    return fp


def kind_of(value) -> Optional[str]:
    """"block" / "st" when `value` (a bound method being assigned to `forward`) is one of sample.py's two functions, else None -- with
    one warning per function when the NAME matches and the body does not (the function is then installed and runs itself)."""
    fn = getattr(value, "__func__", None)
    kind = _NAMES.get(getattr(fn, "__name__", None))
    if kind is None or getattr(fn, "__code__", None) is None:
        return None
    if fn.__code__ not in _fp_cache:
        _fp_cache[fn.__code__] = fingerprint(fn)
    fp = _fp_cache[fn.__code__]
    if fp in KNOWN[kind]:
        return kind
    key = (fn.__code__.co_filename, fn.__code__.co_firstlineno)
    if key not in _warned:
        _warned.add(key)
        why = "its source cannot be read" if fp is None else "its body differs from sample.py's"
        warnings.warn(f"{fn.__name__} ({key[0]}:{key[1]}) is named like sample.py's patched forward but {why}: it is installed as "
                      "written and runs on the strict module route (cd360.sample_py_patch.trust() declares it equivalent)", stacklevel=3)
    return None
