"""MI355X-native CLIP / CLIP+FDT contrastive training path."""
__version__ = "0.1.0"
