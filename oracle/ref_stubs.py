"""Stub loader that lets the reference solver modules import in THIS container.

TEST INFRASTRUCTURE ONLY.  Used solely by ``oracle/make_golden.py`` (fixture
generation, build container only) -- /root/reference does not exist on the GPU
box and nothing in the product, the gpu tests, smoke() or bench.py imports
this file.

The reference's ``scheduler_ppo.py`` (lines 19-23) and
``edit_ppo/scheduler_fmppo.py`` (lines 22-27) import five names from the
un-vendored ``diffusers`` package and one module
(``factor_net_ppo_continous``) that is missing from the reference tree.  We
inject minimal stand-ins for those *names* into ``sys.modules`` -- no
arithmetic lives in them -- and import the reference read-only from where it
lies.  No reference source is copied.
"""
import enum
import functools
import inspect
import io
import contextlib
import sys
import types

REF_ROOT = "/root/reference"


class _Config(dict):
    """attribute-and-.get access, like diffusers' FrozenDict."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def _register_to_config(init):
    sig = inspect.signature(init)

    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = _Config({k: v for k, v in bound.arguments.items() if k != "self"})
        self.config = cfg
        init(self, *args, **kwargs)
    return wrapper


class _ConfigMixin:
    @classmethod
    def from_pretrained(cls, *a, **kw):  # generate_ours.py:127 style
        kw.pop("subfolder", None)
        return cls(**kw)


class _SchedulerMixin:
    pass


class _Karras(enum.Enum):
    DDIMScheduler = 1


class _SchedulerOutput(dict):
    def __init__(self, **kw):
        super().__init__(**kw)
        self.__dict__.update(kw)


class _Logger:
    def __getattr__(self, name):
        return lambda *a, **k: None


def install(flavour):
    """flavour: 'sd' (top-level dir) or 'flux' (edit_ppo). One per process."""
    assert flavour in ("sd", "flux")
    sys.dont_write_bytecode = True
    d = types.ModuleType("diffusers")
    cu = types.ModuleType("diffusers.configuration_utils")
    cu.ConfigMixin = _ConfigMixin
    cu.register_to_config = _register_to_config
    sch = types.ModuleType("diffusers.schedulers")
    su = types.ModuleType("diffusers.schedulers.scheduling_utils")
    su.SchedulerMixin = _SchedulerMixin
    su.KarrasDiffusionSchedulers = _Karras
    su.SchedulerOutput = _SchedulerOutput
    ut = types.ModuleType("diffusers.utils")
    ut.BaseOutput = _SchedulerOutput
    ut.is_scipy_available = lambda: True
    lg = types.ModuleType("diffusers.utils.logging")
    lg.get_logger = lambda *a, **k: _Logger()
    ut.logging = lg
    d.configuration_utils = cu
    d.schedulers = sch
    d.utils = ut
    sch.scheduling_utils = su
    cont = types.ModuleType("factor_net_ppo_continous")
    cont.FactorNetPPOContinous = object
    for name, mod in {
        "diffusers": d, "diffusers.configuration_utils": cu,
        "diffusers.schedulers": sch, "diffusers.schedulers.scheduling_utils": su,
        "diffusers.utils": ut, "diffusers.utils.logging": lg,
        "factor_net_ppo_continous": cont,
    }.items():
        sys.modules[name] = mod
    sys.path.insert(0, REF_ROOT if flavour == "sd" else REF_ROOT + "/edit_ppo")


@contextlib.contextmanager
def quiet():
    """The reference prints every step (scheduler_ppo.py:243,289)."""
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        yield buf
