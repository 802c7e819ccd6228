"""Causal text transformer container + tokenisation front end
(reference prototype/model/text_encoder/text_transformer.py:21-368, 'Transformer' branch only)."""
import torch
from torch import nn

from ..image_encoder.base_transformer import Transformer, LayerNorm, init_blocks

VOCAB_SIZE = 49409          # OpenAI BPE (49408) + the reference's extra <|mask|> token
SOT, EOT = 49407, 49408


class TokenizedOutput:
    def __init__(self, output):
        self.out1, self.out2 = output

    def cuda(self):
        return self.out1.cuda(), self.out2.cuda()


class TextTransformer(nn.Module):
    def __init__(self, embed_dim, context_length, transformer_width, transformer_heads, transformer_layers,
                 positional_embedding_flag=True, checkpoint=False, bpe_path=None, text_encode_type="Transformer",
                 text_model_utils=None):
        super().__init__()
        if text_encode_type != "Transformer":
            raise NotImplementedError("text_encode_type=%r: only the 'Transformer' branch is on the hot path" % text_encode_type)
        if not positional_embedding_flag:
            raise NotImplementedError("positional_embedding_flag=False is not used by any shipped config")
        self.context_length = context_length
        self.positional_embedding_flag = positional_embedding_flag
        self.text_encode_type = text_encode_type
        self.text_model_utils = text_model_utils or {}
        self.bpe_path = bpe_path
        self._tokenizer = None
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads)
        self.vocab_size = VOCAB_SIZE
        self.token_embedding = nn.Embedding(self.vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Linear(transformer_width, embed_dim)
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        init_blocks(self.transformer)
        nn.init.normal_(self.text_projection.weight, std=transformer_width ** -0.5)

    @property
    def tokenizer(self):
        if self._tokenizer is None:
            from ..utils.text_utils.simple_tokenizer import SimpleTokenizer
            self._tokenizer = SimpleTokenizer(self.bpe_path)
        return self._tokenizer

    @property
    def native_tokenizer(self):
        if getattr(self, "_native_tokenizer", None) is None:
            from ..utils.text_utils.simple_tokenizer import NativeTokenizer
            self._native_tokenizer = NativeTokenizer(self.tokenizer)
        return self._native_tokenizer

    def tokenize(self, texts, context_length=None, return_length=False, mask_type=None):
        """list[str] -> (tokens int64 [B,ctx], pad_mask fp32 [B,ctx] with 0 valid / -inf pad); over-long captions keep
        [sot] + tok[1:ctx-1] + [eot] (reference text_transformer.py:155-202)."""
        if mask_type is not None:
            raise NotImplementedError("MLM masking is not on the contrastive hot path")
        ctx = context_length or self.context_length
        if isinstance(texts, str):
            texts = [texts]
        # C++ BPE (ilvlm_tokenizer_*): the reference runs this loop in Python inside every forward()
        result, pad_mask, lengths = self.native_tokenizer.encode_batch(list(texts), ctx, SOT, EOT)
        if return_length:
            return result, lengths, pad_mask
        return result, pad_mask

    def wrap_tokenize(self, text):
        return TokenizedOutput(self.tokenize(text))

    def forward(self, text, mask_type=None, return_dense=False, return_raw_feature=False, return_padmask=False,
                return_att=False, raw_text=True):
        """Inference-only call with the reference signature (text_transformer.py:211-338): [projected EOT feature,
        ln_final word features, EOT feature, pad mask] as requested.  Training goes through the owning model."""
        if mask_type is not None or return_att:
            raise NotImplementedError("MLM masking / attention maps are not on the HIP path")
        owner = getattr(self, "_owner", lambda: None)()
        if owner is None:
            raise RuntimeError("TextTransformer runs inside a CLIP / Clip_FDT model (its engine owns the kernels)")
        with torch.no_grad():
            e = owner._eng
            e.prepare()
            tokens, pad_mask = owner._text_inputs(text if raw_text else tuple(text), e.arena.P.device)
            B, Lt = tokens.shape
            xt, _ = e.text_fwd(tokens, False)
            proj, feat, _ = e.text_pooled(xt, tokens, B, Lt, False)
            ret = [proj]
            if return_dense:
                words, _ = e.text_words(xt, False)
                ret.append(words.view(B, Lt, -1))
        if return_raw_feature:
            ret.append(feat)
        if return_padmask:
            ret.append(pad_mask)
        return ret[0] if len(ret) == 1 else ret


def text_transformers(**kwargs):
    d = dict(context_length=77, transformer_width=512, transformer_heads=8, transformer_layers=12,
             positional_embedding_flag=True, checkpoint=False)
    d.update(kwargs)
    return TextTransformer(**d)


def text_transformers_L(**kwargs):
    d = dict(context_length=77, transformer_layers=12, transformer_width=768, transformer_heads=12,
             positional_embedding_flag=True, checkpoint=False)
    d.update(kwargs)
    return TextTransformer(**d)
