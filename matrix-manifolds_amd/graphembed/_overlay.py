"""Overlay of this package onto a maintainer's checkout of the reference (option A).

This package defines the hot path only (`manifolds`, `modules`, `optim`, `objectives`, `data`, `pyx`, `metrics`,
`utils`, `linalg.fast`).  The reference's control plane — `graphembed.train`, `.train_da`, `.products`, `.monitor`,
`.linalg.torch_batch`, `.inference` (run.py:15-18, 76-81; `graphembed/__init__.py:1-9`) — is NOT rebuilt here.
(`graphembed.linalg` has no `__init__.py` on either side: once the checkout's directory is on `__path__` it is a namespace
package over both — `fast` from here, `torch_batch` from the checkout.)  When another `graphembed`
package follows this one on `sys.path` (the maintainer's checkout), `install()`

  * appends its directories to `__path__` of this package and of its sub-packages, so every sub-module this package does
    not define (`graphembed.train`, `graphembed.products.embedding`, `graphembed.manifolds.universal`, …) is imported
    from the checkout, while the names defined here keep shadowing the reference's, and
  * gives the modules defined here a module-level `__getattr__` (PEP 562) that resolves a name they do NOT define
    (`graphembed.manifolds.Universal`, `graphembed.objectives.KLDiveregenceLoss`, `graphembed.utils.PLT_MUTEX`,
    `graphembed.modules.EmbeddingBase`, …) in the checkout's counterpart.

Nothing of the reference is copied or shipped: without a second `graphembed` on the path (the GPU box, the tests) this
module does nothing.
"""
import importlib
import importlib.util
import os
import re
import sys

_OURS = os.path.dirname(os.path.abspath(__file__))
_ROOT = 'graphembed'
_SUBPACKAGES = ('manifolds', 'optim', 'data', 'pyx')
_MODULES = ('modules', 'objectives', 'utils', 'metrics')
_loaded = {}


_later_cache = {}


def later_packages():
    """Directories of other `graphembed` packages on sys.path, in path order (this one excluded).  Cached per state of
    sys.path: an attribute miss on an overlaid module must not rescan the path every time."""
    key = tuple(sys.path)
    hit = _later_cache.get(key)
    if hit is None:
        _later_cache.clear()
        hit = _later_cache[key] = tuple(_scan_later_packages())
    return list(hit)


def _scan_later_packages():
    out = []
    for p in sys.path:
        cand = os.path.join(p or os.getcwd(), _ROOT)
        try:
            if not os.path.isfile(os.path.join(cand, '__init__.py')) or os.path.samefile(cand, _OURS):
                continue
        except OSError:
            continue
        if cand not in out:
            out.append(cand)
    return out


def _counterpart(stem):
    """The checkout's `<stem>.py`, loaded once under a private name (its `from graphembed... import` lines see the
    merged package)."""
    if stem in _loaded:
        return _loaded[stem]
    mod = None
    for base in later_packages():
        path = os.path.join(base, stem + '.py')
        if os.path.isfile(path):
            name = f'{_ROOT}._overlaid_{stem}'
            spec = importlib.util.spec_from_file_location(name, path)
            mod = importlib.util.module_from_spec(spec)
            mod.__package__ = _ROOT
            sys.modules[name] = mod
            try:
                spec.loader.exec_module(mod)
            except BaseException:
                sys.modules.pop(name, None)
                raise
            break
    _loaded[stem] = mod
    return mod


def _module_getattr(stem):
    def __getattr__(name):
        if not name.startswith('__'):
            try:
                mod = _counterpart(stem)
            except Exception as e:  # noqa: BLE001 — the checkout's counterpart does not import here: a miss, with the cause
                raise AttributeError(f"module '{_ROOT}.{stem}' has no attribute '{name}' (the checkout's {stem}.py does not "
                                     f"import: {type(e).__name__}: {e})") from e
            if mod is not None and hasattr(mod, name):
                return getattr(mod, name)
        raise AttributeError(f"module '{_ROOT}.{stem}' has no attribute '{name}'")
    return __getattr__


def _defines(path, name):
    """True when the source file binds `name` at its top level (def / class / assignment / import ... as): a textual test
    that keeps an attribute miss (`hasattr(graphembed, 'torch')`, pytest / inspect / pickle probes) from importing every
    module of the checkout that merely MENTIONS the name."""
    pat = re.compile(r'^(?:(?:async\s+)?def\s+{0}\b|class\s+{0}\b|{0}\s*(?::[^=\n]+)?=[^=]|'
                     r'(?:from\s+\S+\s+)?import\s+.*\b{0}\b\s*(?:,|$|#)|.*\bas\s+{0}\b)'.format(re.escape(name)), re.M)
    try:
        with open(path, encoding='utf-8', errors='replace') as f:
            return pat.search(f.read()) is not None
    except OSError:
        return False


def _is_submodule(modname, name):
    """Is `modname.name` a module file or package directory on the merged __path__?  (Nothing is imported to find out.)"""
    mod = sys.modules.get(modname)
    for d in list(getattr(mod, '__path__', ())):
        if os.path.isfile(os.path.join(d, name + '.py')) or os.path.isdir(os.path.join(d, name)):
            return True
    return False


def _package_getattr(modname, rel):
    def __getattr__(name):
        if name.startswith('__'):
            raise AttributeError(name)
        full = f'{modname}.{name}'
        if _is_submodule(modname, name):       # a sub-module / sub-package of the checkout (graphembed.inference, data.preprocess)
            try:
                return importlib.import_module(full)
            except ModuleNotFoundError as e:
                if e.name != full:
                    # the checkout's module exists but one of ITS imports is missing here (train.py: tensorboard): that is the
                    # caller's error when the module was asked for by name — and stays an AttributeError for getattr-with-default
                    # probes, with the cause attached
                    raise AttributeError(f"module '{modname}' has no usable attribute '{name}': {e}") from e
        ours = os.path.join(_OURS, rel)
        for base in later_packages():          # a name one of the checkout's sub-modules defines (manifolds.Universal)
            d = os.path.join(base, rel)
            if not os.path.isdir(d):
                continue
            for fn in sorted(os.listdir(d)):
                stem, ext = os.path.splitext(fn)
                if ext != '.py' or stem == '__init__' or os.path.exists(os.path.join(ours, fn)):
                    continue
                if not _defines(os.path.join(d, fn), name):   # (only a module that BINDS the name is imported)
                    continue
                try:
                    sub = importlib.import_module(f'{modname}.{stem}')
                except Exception as e:  # noqa: BLE001 — a speculative import must not turn an attribute miss into its error
                    raise AttributeError(f"module '{modname}' has no attribute '{name}' "
                                         f"({modname}.{stem} defines it but does not import: {type(e).__name__}: {e})") from e
                if hasattr(sub, name):
                    return getattr(sub, name)
        raise AttributeError(f"module '{modname}' has no attribute '{name}'")
    return __getattr__


def install():
    """Idempotent; a no-op unless another `graphembed` package follows this one on sys.path."""
    later = later_packages()
    if not later:
        return False
    root = sys.modules[_ROOT]
    for base in later:
        if base not in root.__path__:
            root.__path__.append(base)
    root.__dict__.setdefault('__getattr__', _package_getattr(_ROOT, ''))
    for sub in _SUBPACKAGES:
        mod = sys.modules.get(f'{_ROOT}.{sub}')
        if mod is None:
            continue
        for base in later:
            d = os.path.join(base, sub)
            if os.path.isdir(d) and d not in mod.__path__:
                mod.__path__.append(d)
        mod.__dict__.setdefault('__getattr__', _package_getattr(f'{_ROOT}.{sub}', sub))
    for stem in _MODULES:
        mod = sys.modules.get(f'{_ROOT}.{stem}')
        if mod is not None:
            mod.__dict__.setdefault('__getattr__', _module_getattr(stem))
    return True
