from .scheduler import Cosine, CosineLRScheduler  # noqa: F401


def scheduler_entry(config):
    """reference prototype/lr_scheduler/__init__.py:18-22"""
    return globals()[config["type"]](**config["kwargs"])
