"""Evaluation-side loader with the interface CLIP_benchmark's custom model type expects
(reference CLIP_benchmark/clip_benchmark/models/fdt.py:16-91): builds the model from the training YAML, loads one
checkpoint or the element-wise average of several ('module.' prefix stripped), and exposes encode_image / encode_text /
get_tokenize_function.  Encoders run the no-grad HIP path (no activations are saved)."""
import torch

from .prototype.model import model_entry
from .prototype.utils.misc import count_params, load_state_model, parse_config


def _strip(state):
    return {(k[7:] if k.startswith("module.") else k): v for k, v in state.items()}


def average_checkpoints(paths):
    """{key: mean over checkpoints} of the 'model' entries (fdt.py:28-40)."""
    acc, n = {}, 0
    for path in paths:
        state = _strip(torch.load(path, map_location="cpu", weights_only=False)["model"])
        for k, v in state.items():
            acc[k] = v.clone().float() if k not in acc else acc[k] + v.float()
        n += 1
    return {k: v / n for k, v in acc.items()}


class MyModelZoo(torch.nn.Module):
    def __init__(self, config, ckpt_pth=None):
        super().__init__()
        cfg = parse_config(config) if isinstance(config, str) else config
        self.model = model_entry(cfg.model if hasattr(cfg, "model") else cfg["model"])
        self.model.cuda()
        count_params(self.model)
        if isinstance(ckpt_pth, (list, tuple)):
            load_state_model(self.model, average_checkpoints(ckpt_pth))
        elif ckpt_pth:
            load_state_model(self.model, _strip(torch.load(ckpt_pth, map_location="cpu", weights_only=False)["model"]))
        self.model.eval()
        self.is_fdt = hasattr(self.model, "extract_img_sd_ft")

    @torch.no_grad()
    def encode_image(self, image):
        image = image.cuda()
        if self.is_fdt:
            return self.model.extract_img_sd_ft(image)[1]
        return self.model.encode_image(image)

    @torch.no_grad()
    def encode_text(self, text_tokenize):
        if self.is_fdt:
            return self.model.extract_txt_sd_ft(text_tokenize, raw_text=True)[1]
        return self.model.encode_text(text_tokenize, raw_text=True)

    @torch.no_grad()
    def get_full_image_embedding_info(self, image):
        return self.model.extract_img_sd_ft(image)

    def get_tokenize_function(self):
        return self.model.encode_text.wrap_tokenize

    def get_test_transform(self):
        raise NotImplementedError("image preprocessing (torchvision ONECROP) belongs to the input pipeline, outside the hot path")


def load_fdt(model_name, pretrained, cache_dir=None, device="cuda"):
    config = "example/clip/config_cc3m.yaml" if model_name == "clip" else "example/clip_fdt/config_cc3m.yaml"
    return MyModelZoo(config=config, ckpt_pth=pretrained), None, None
