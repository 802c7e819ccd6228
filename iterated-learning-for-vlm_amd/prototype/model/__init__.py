"""String-keyed model registry (reference prototype/model/__init__.py:1-6)."""
from .clip_fdt import clip_fdt_vitb16, clip_fdt_vitb32, clip_fdt_vitL14  # noqa: F401
from .clip import clip_vitb32  # noqa: F401


def model_entry(config):
    return globals()[config["type"]](**config["kwargs"])
