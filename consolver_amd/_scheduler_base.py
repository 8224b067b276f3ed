"""Scheduler base classes: diffusers' own mixins when diffusers is importable, a stand-alone equivalent otherwise.

The reference's schedulers are ``class PPOScheduler(SchedulerMixin, ConfigMixin)`` with a ``@register_to_config``
constructor (scheduler_ppo.py:19-20,48,81-97; edit_ppo/scheduler_fmppo.py:22-24,56,107) and are handed to
``StableDiffusionPipeline.from_pretrained(path, scheduler=noise_scheduler, ...)`` (gen_ppo.py:173-179), which accepts a
passed component only if it is an instance of the class family named in ``model_index.json`` (for schedulers:
``issubclass(type(scheduler), SchedulerMixin)``) and then probes ``scheduler.config.steps_offset`` /
``scheduler.config.clip_sample``, replacing ``scheduler._internal_dict`` when ``steps_offset != 1``.  So:

* with diffusers installed, ``SchedulerMixin`` / ``ConfigMixin`` / ``register_to_config`` / ``KarrasDiffusionSchedulers``
  below ARE diffusers' objects and our schedulers subclass them (``save_config`` / ``from_config`` / ``save_pretrained`` /
  ``from_pretrained`` / ``compatibles`` come from diffusers);
* without it (this image), the classes below provide the same surface on the same file format
  (``scheduler_config.json`` with ``_class_name``), so a scheduler saved by either loads with the other.

The choice is made once, at import time (class bases cannot change later).
"""
import functools
import inspect
import json
import os

import numpy as np

try:                                                    # pragma: no cover - exercised by tests/test_diffusers_dropin.py
    from diffusers.configuration_utils import ConfigMixin, register_to_config
    from diffusers.schedulers.scheduling_utils import SchedulerMixin
    try:
        from diffusers.schedulers.scheduling_utils import KarrasDiffusionSchedulers
        KARRAS_COMPATIBLES = [e.name for e in KarrasDiffusionSchedulers]
    except ImportError:
        KARRAS_COMPATIBLES = None
    HAVE_DIFFUSERS = True
except ImportError:
    HAVE_DIFFUSERS = False
    KARRAS_COMPATIBLES = None

if KARRAS_COMPATIBLES is None:
    # the members of diffusers 0.26.3's KarrasDiffusionSchedulers enum (env.yaml:52), by name
    KARRAS_COMPATIBLES = [
        "DDIMScheduler", "DDPMScheduler", "PNDMScheduler", "LMSDiscreteScheduler", "EulerDiscreteScheduler",
        "HeunDiscreteScheduler", "EulerAncestralDiscreteScheduler", "DPMSolverMultistepScheduler",
        "DPMSolverSinglestepScheduler", "KDPM2DiscreteScheduler", "KDPM2AncestralDiscreteScheduler",
        "DEISMultistepScheduler", "UniPCMultistepScheduler", "DPMSolverSDEScheduler", "EDMEulerScheduler",
    ]


class SolverConfig(dict):
    """read-only mapping with attribute *and* ``.get`` access, like diffusers' FrozenDict
    (edit_ppo/pipeline.py:1013-1016 uses ``scheduler.config.get``)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def _frozen(self, *a, **k):
        raise TypeError("scheduler configs are immutable; use from_config(config, key=value)")

    __setitem__ = __delitem__ = update = pop = popitem = setdefault = clear = _frozen
    __setattr__ = _frozen

    # frozen, but copyable and picklable like diffusers' FrozenDict (copy.deepcopy(scheduler), torch.save, spawned workers): rebuilt from a plain dict
    def __reduce__(self):
        return (SolverConfig, (dict(self),))

    def __copy__(self):
        return SolverConfig(dict(self))

    def __deepcopy__(self, memo):
        import copy
        return SolverConfig(copy.deepcopy(dict(self), memo))


if not HAVE_DIFFUSERS:
    SCHEDULER_CONFIG_NAME = "scheduler_config.json"

    def register_to_config(init):
        """records the constructor's arguments (defaults applied) in ``self.config`` BEFORE the body runs; keyword
        arguments that start with an underscore go to the config only (what ``from_config`` passes back)."""
        sig = inspect.signature(init)

        @functools.wraps(init)
        def inner_init(self, *args, **kwargs):
            init_kwargs = {k: v for k, v in kwargs.items() if not k.startswith("_")}
            hidden = {k: v for k, v in kwargs.items() if k.startswith("_")}
            if not isinstance(self, ConfigMixin):
                raise RuntimeError(f"`@register_to_config` was applied to {type(self).__name__}, which does not inherit from `ConfigMixin`.")
            bound = sig.bind(self, *args, **init_kwargs)
            given = set(bound.arguments) - {"self"}
            bound.apply_defaults()
            new = {k: v for k, v in bound.arguments.items() if k != "self"}
            defaults_used = sorted(set(new) - given)
            if defaults_used:
                new["_use_default_values"] = defaults_used
            self.register_to_config(**{**hidden, **new})
            init(self, *args, **init_kwargs)
        return inner_init

    class ConfigMixin:
        config_name = None
        ignore_for_config = []
        has_compatibles = False

        def register_to_config(self, **kwargs):
            if self.config_name is None:
                raise NotImplementedError(f"Make sure that {self.__class__} has defined a class name `config_name`")
            kwargs.pop("kwargs", None)
            prev = dict(getattr(self, "_internal_dict", {}))
            prev.update(kwargs)
            self._internal_dict = SolverConfig(prev)

        @property
        def config(self):
            return self._internal_dict

        # ---- serialisation: <dir>/scheduler_config.json, the diffusers layout -------------------------------------
        def to_json_string(self):
            def saveable(v):
                if isinstance(v, np.ndarray):
                    return v.tolist()
                if isinstance(v, (np.integer, np.floating)):
                    return v.item()
                if isinstance(v, dict):
                    return {k: saveable(x) for k, x in v.items()}
                if isinstance(v, (list, tuple)):
                    return [saveable(x) for x in v]
                return v
            d = {k: saveable(v) for k, v in dict(self.config).items() if k != "_use_default_values"}
            d["_class_name"] = self.__class__.__name__
            from . import __version__
            d["_consolver_amd_version"] = __version__
            return json.dumps(d, indent=2, sort_keys=True) + "\n"

        def save_config(self, save_directory, push_to_hub=False, **kwargs):
            if push_to_hub:
                raise ValueError("push_to_hub needs diffusers / a network; save locally instead")
            if os.path.isfile(save_directory):
                raise AssertionError(f"Provided path ({save_directory}) should be a directory, not a file")
            os.makedirs(save_directory, exist_ok=True)
            with open(os.path.join(save_directory, self.config_name), "w", encoding="utf-8") as f:
                f.write(self.to_json_string())

        @classmethod
        def load_config(cls, pretrained_model_name_or_path, return_unused_kwargs=False, subfolder=None, **kwargs):
            path = str(pretrained_model_name_or_path)
            if os.path.isfile(path):
                cfg_file = path
            else:
                d = os.path.join(path, subfolder) if subfolder else path
                cfg_file = os.path.join(d, cls.config_name)
            if not os.path.isfile(cfg_file):
                raise EnvironmentError(f"{path!r} is not a local directory holding {cls.config_name} (hub ids need diffusers and a network)")
            with open(cfg_file, "r", encoding="utf-8") as f:
                cfg = json.load(f)
            return (cfg, kwargs) if return_unused_kwargs else cfg

        @classmethod
        def extract_init_dict(cls, config_dict, **kwargs):
            params = [n for i, n in enumerate(inspect.signature(cls.__init__).parameters) if i > 0 and n not in ("kwargs", "args")]
            cfg = {k: v for k, v in dict(config_dict).items() if k != "_use_default_values"}
            init_dict = {}
            for k in params:
                if k in kwargs:
                    init_dict[k] = kwargs.pop(k)
                elif k in cfg:
                    init_dict[k] = cfg.pop(k)
            hidden = {k: v for k, v in cfg.items() if k.startswith("_") and k != "_class_name"}
            unused = {**{k: v for k, v in cfg.items() if not k.startswith("_")}, **kwargs}
            return init_dict, unused, hidden

        @classmethod
        def from_config(cls, config=None, return_unused_kwargs=False, **kwargs):
            if config is None:
                raise ValueError("Please make sure to provide a config as the first positional argument.")
            if not isinstance(config, dict):
                raise ValueError("`config` must be a dict (a scheduler's .config); use from_pretrained for a path")
            init_dict, unused, hidden = cls.extract_init_dict(config, **kwargs)
            obj = cls(**init_dict)
            if hidden:
                obj.register_to_config(**hidden)
            return (obj, unused) if return_unused_kwargs else obj

    class SchedulerMixin:
        config_name = SCHEDULER_CONFIG_NAME
        _compatibles = []
        has_compatibles = True

        @classmethod
        def from_pretrained(cls, pretrained_model_name_or_path=None, subfolder=None, return_unused_kwargs=False, **kwargs):
            config, kwargs = cls.load_config(pretrained_model_name_or_path, subfolder=subfolder, return_unused_kwargs=True, **kwargs)
            return cls.from_config(config, return_unused_kwargs=return_unused_kwargs, **kwargs)

        def save_pretrained(self, save_directory, push_to_hub=False, **kwargs):
            self.save_config(save_directory=save_directory, push_to_hub=push_to_hub, **kwargs)

        @property
        def compatibles(self):
            return self._get_compatibles()

        @classmethod
        def _get_compatibles(cls):
            import consolver_amd
            names = list(set([cls.__name__] + list(cls._compatibles)))
            return [getattr(consolver_amd, n) for n in names if hasattr(consolver_amd, n)]
