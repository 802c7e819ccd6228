from .loss import ClipInfoCELoss  # noqa: F401
