"""Zero-edit binding of the fused fast path into an UNCHANGED Free-SurGS checkout ("no edit, one
environment variable").

The reference's driver picks its hot-path functions up BY NAME at import time:

    train.py:5     from gaussian_renderer import render, render_custom, inference
    train.py:30    from utils.loss_utils import rgb_loss_func, pearson_depth_loss, local_pearson_loss
    scene/gaussian_model.py:378,405   self.optimizer = torch.optim.Adam(...)
    scene/pose_optimizer.py:490       self.optimizer = optim.Adam([...])            (import torch.optim as optim, :3)
    scene/pose_optimizer.py:888,890   torch.optim.Adam(...)

so rebinding those names right after the defining module has executed -- and before train.py copies them into its own
namespace -- routes the unchanged driver through `fsgs_amd.render.render` (one fused HIP op with the signature and the
10-key dict of gaussian_renderer/__init__.py:49-92), the fused loss kernels (`fsgs_amd.losses`, same signatures as
utils/loss_utils.py:47-54,98-127) and `fsgs_amd.optim.FusedAdam` (the arguments of torch.optim.Adam) with no source edit.

How: `install()` puts a finder in front of `sys.meta_path` that lets the normal machinery find a target module and wraps
its loader's `exec_module`: the patch runs when the module body has finished.  A target that is ALREADY executing when
install() runs cannot be caught that way (its body will still define the name and overwrite whatever is put there now);
for it the module object's class is swapped for a subclass whose data descriptor serves the replacement to every later
`getattr` / `from ... import` -- which is how train.py:5 reads `render`.  That case is the normal one:
`FSGS_AUTOBIND=1` makes the import-name shims (`diff_gaussian_rasterization`, `simple_knn`) call install(), and the
first of them is imported from the middle of gaussian_renderer/__init__.py (:15).

Only names are rebound; nothing is monkey-patched inside torch, and a module that is not one of the four targets is
never touched.  `uninstall()` removes the finder (bindings already made stay).  `bound()` reports what was rebound.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

ENV = "FSGS_AUTOBIND"


# ---- what is bound where --------------------------------------------------------------------------------------------
def _render():
    from .render import render

    return render


def _loss(name):
    def get():
        from . import losses

        return getattr(losses, name)

    return get


def _fused_adam():
    from .optim import FusedAdam

    return FusedAdam


class _OptimProxy(types.ModuleType):
    """`torch.optim` with `Adam` = FusedAdam; every other attribute is torch.optim's own"""

    def __init__(self):
        super().__init__("torch.optim")
        import torch.optim

        self.__dict__["_real"] = torch.optim

    @property
    def Adam(self):
        return _fused_adam()

    def __getattr__(self, name):
        return getattr(self.__dict__["_real"], name)


class _TorchProxy(types.ModuleType):
    """the module-global `torch` of scene/gaussian_model.py / scene/pose_optimizer.py: `torch.optim.Adam` = FusedAdam,
    everything else is the real torch (attribute lookups are forwarded, nothing inside torch is modified)"""

    def __init__(self):
        super().__init__("torch")
        import torch

        self.__dict__["_real"] = torch
        self.__dict__["optim"] = _OptimProxy()

    def __getattr__(self, name):
        return getattr(self.__dict__["_real"], name)


def _is_real(mod_dict, name, real_name):
    m = mod_dict.get(name)
    return isinstance(m, types.ModuleType) and getattr(m, "__name__", None) == real_name and not isinstance(
        m, (_TorchProxy, _OptimProxy))


def _add_densification_stats(self, viewspace_point_tensor, update_filter):
    """GaussianModel.add_densification_stats (scene/gaussian_model.py:678-681) without the boolean-mask indexing: the same
    sums, `where` instead of `x[mask] += ...`, so no host synchronisation (two of them per iteration in the original)"""
    import torch

    g = viewspace_point_tensor.grad
    f = update_filter.reshape(-1, 1)
    norm = torch.norm(g, dim=-1, keepdim=True)  # (all three components, as the reference; the third is zero)
    self.variables["xyz_gradient_accum"] += torch.where(f, norm, torch.zeros_like(norm))
    self.variables["denom"] += f.to(self.variables["denom"].dtype)


def _bind_adam(mod):
    """the two spellings the reference uses: `torch.optim.Adam(...)` and `optim.Adam(...)`; in scene.gaussian_model also the
    sync-free statement of GaussianModel.add_densification_stats (once the class exists: a no-op on an in-flight module)"""
    done = []
    d = mod.__dict__
    cls = d.get("GaussianModel")
    if isinstance(cls, type) and "add_densification_stats" in vars(cls) and not hasattr(cls, "_fsgs_original_add_densification_stats"):
        cls._fsgs_original_add_densification_stats = cls.add_densification_stats
        cls.add_densification_stats = _add_densification_stats
        done.append("GaussianModel.add_densification_stats")
    if _is_real(d, "torch", "torch"):
        d["torch"] = _TorchProxy()
        done.append("torch.optim.Adam")
    if _is_real(d, "optim", "torch.optim"):
        d["optim"] = _OptimProxy()
        done.append("optim.Adam")
    return done


# module name -> {attribute: getter of the replacement}, or a callable(module) -> list of what it rebound
TARGETS = {
    "gaussian_renderer": {"render": _render},
    "utils.loss_utils": {"rgb_loss_func": _loss("rgb_loss_func"), "pearson_depth_loss": _loss("pearson_depth_loss"),
                         "local_pearson_loss": _loss("local_pearson_loss")},
    "scene.gaussian_model": _bind_adam,
    "scene.pose_optimizer": _bind_adam,
}

_bound = {}  # module name -> list of the names rebound in it
_pending = []  # in-flight modules to look at again once their body has finished


def bound():
    return {k: list(v) for k, v in _bound.items()}


def _patch_finished(mod):
    spec = TARGETS[mod.__name__]
    if callable(spec):
        done = spec(mod)
    else:
        done = []
        for attr, get in spec.items():
            # only names the module really defines (a stand-in without them is left alone), and only once
            if attr in mod.__dict__ and "_fsgs_original_" + attr not in mod.__dict__:
                mod.__dict__["_fsgs_original_" + attr] = mod.__dict__[attr]
                mod.__dict__[attr] = get()
                done.append(attr)
    if done:
        _bound.setdefault(mod.__name__, [])
        _bound[mod.__name__] += [d for d in done if d not in _bound[mod.__name__]]


def _patch_in_flight(mod):
    """a target whose body is still running: serve the replacements through a data descriptor on a subclass of the
    module's class (class-level data descriptors win over the instance dict in attribute lookup)"""
    spec = TARGETS[mod.__name__]
    if callable(spec):
        # the optimizer proxies replace module GLOBALS (`torch`, `optim`), which the reference imports at the very top
        # (scene/gaussian_model.py:12, scene/pose_optimizer.py:1,3 -- before the rasteriser import that brings us here): they
        # are in place already and nothing below rebinds them, so patch now; the rest of the body runs on the proxies
        done = spec(mod)
        if done:
            _bound.setdefault(mod.__name__, [])
            _bound[mod.__name__] += [d for d in done if d not in _bound[mod.__name__]]
        _pending.append(mod)  # what the body has not defined yet (the class) is bound when it has finished: _drain_pending
        return bool(done)
    if not type(mod).__name__.startswith("_FsgsBound_"):  # install() runs from both import shims: one swap, not a stack of them
        ns = {}
        for attr, get in spec.items():
            ns[attr] = property(lambda self, _g=get: _g())
        mod.__class__ = type("_FsgsBound_" + mod.__name__.replace(".", "_"), (type(mod),), ns)
    _bound.setdefault(mod.__name__, [])
    _bound[mod.__name__] += [a for a in spec if a not in _bound[mod.__name__]]
    return True


class _Loader(importlib.abc.Loader):
    def __init__(self, inner):
        self.inner = inner

    def create_module(self, spec):
        return self.inner.create_module(spec) if hasattr(self.inner, "create_module") else None

    def exec_module(self, module):
        self.inner.exec_module(module)
        _patch_finished(module)

    def __getattr__(self, name):  # get_source, get_filename, is_package, ... of the real loader
        return getattr(self.inner, name)


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        if _pending:
            _drain_pending()
        if fullname not in TARGETS:
            return None
        for f in sys.meta_path:
            if f is self or not hasattr(f, "find_spec"):
                continue
            spec = f.find_spec(fullname, path, target)
            if spec is not None:
                if spec.loader is not None and hasattr(spec.loader, "exec_module"):
                    spec.loader = _Loader(spec.loader)
                return spec
        return None


_finder = None


def _executing(mod):
    spec = getattr(mod, "__spec__", None)
    return bool(spec is not None and getattr(spec, "_initializing", False))


def _drain_pending():
    """every import passes the finder: a module that was in flight at install() is finished off at the first import after
    its body has run (also callable directly)"""
    for mod in list(_pending):
        if not _executing(mod):
            _pending.remove(mod)
            _patch_finished(mod)


def install():
    """idempotent.  -> {module: how it was handled} for the targets that were already in sys.modules"""
    global _finder
    report = {}
    if _finder is None:
        _finder = _Finder()
        sys.meta_path.insert(0, _finder)
    for name in TARGETS:
        mod = sys.modules.get(name)
        if mod is None:
            continue
        if _executing(mod):
            report[name] = "in flight: class swap" if _patch_in_flight(mod) else "in flight: NOT bound (import fsgs_amd.autobind earlier)"
        else:
            _patch_finished(mod)
            report[name] = "already imported: patched in place (importers that copied the names earlier keep the originals)"
    return report


def uninstall():
    global _finder
    if _finder is not None and _finder in sys.meta_path:
        sys.meta_path.remove(_finder)
    _finder = None


def verify(strict=False):
    """FSGS_AUTOBIND asked for the fused path: every target that HAS been imported and has finished its body must carry its
    bindings.  In-flight detection reads the private ModuleSpec._initializing; should a future CPython rename it, install()
    would take the 'already imported' branch on a module whose body has not defined `render` yet, bind nothing, and train.py
    would run the original with no error.  -> list of problems (warned about loudly; raised when `strict`)."""
    import warnings

    problems = []
    for name, spec in TARGETS.items():
        mod = sys.modules.get(name)
        if mod is None or _executing(mod) or mod in _pending:
            continue
        if callable(spec):
            continue  # (optimizer proxies: bound() reports them; nothing to look up by attribute name)
        for attr in spec:
            if attr in mod.__dict__ and "_fsgs_original_" + attr not in mod.__dict__ and not type(mod).__name__.startswith("_FsgsBound_"):
                problems.append("%s.%s is the ORIGINAL although %s is set" % (name, attr, ENV))
    if problems:
        msg = "fsgs_amd.autobind: " + "; ".join(problems) + " -- import fsgs_amd.autobind before the reference's modules"
        if strict:
            raise RuntimeError(msg)
        warnings.warn(msg, RuntimeWarning, stacklevel=2)
    return problems


def install_from_env():
    """called by the import-name shims: FSGS_AUTOBIND=1 -> install(); what is already importable is verified on the spot and
    everything once more at interpreter exit of the import phase (the finder drains its pending modules on every import)"""
    if os.environ.get(ENV, "") not in ("", "0"):
        rep = install()
        verify()
        return rep
    return None
