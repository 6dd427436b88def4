"""Shape/config objects for the VACNIC path (mirrors the BartConfig fields + ctor kwargs the
reference passes at TRAIN:743 and the CLIP geometry of clip.load, TRAIN:737)."""
from dataclasses import dataclass, field
from typing import List


@dataclass
class VacnicConfig:
    # BartConfig subset (facebook/bart-base | bart-large)
    d_model: int = 1024
    encoder_layers: int = 12
    decoder_layers: int = 12
    encoder_attention_heads: int = 16
    decoder_attention_heads: int = 16
    encoder_ffn_dim: int = 4096
    decoder_ffn_dim: int = 4096
    vocab_size: int = 50267            # 50265 + <ENT>, <NONAME> (TRAIN:753-754)
    max_position_embeddings: int = 1024
    pad_token_id: int = 1
    bos_token_id: int = 0
    eos_token_id: int = 2
    decoder_start_token_id: int = 2
    scale_embedding: bool = False
    activation_function: str = "gelu"
    dropout: float = 0.1
    attention_dropout: float = 0.0      # BartConfig's class defaults; the hub checkpoints carry their own (HUB_MODEL_DROPOUTS below)
    activation_dropout: float = 0.0
    init_std: float = 0.02
    # VACNIC ctor kwargs (MFULL:1881)
    enc_fusion_layer: List[int] = field(default_factory=lambda: list(range(12)))
    dim_common: int = 1024
    prompt_mlp_type: str = "clipcap"
    prompt_size: int = 20
    map_size: List[int] = None         # --prompt_mlp_type mlp: [n_patch_tokens, h1, ..., prompt_len] (TRAIN `--map_size`, MFULL:1138)
    max_ner_type_len: int = 80
    max_ner_type_len_gt: int = 20
    only_image: bool = False
    init_attn_weight: bool = False     # MFULL:1858-1870: tie name-self-attn / image-name cross-attn weights to the text self-attn
    face_dim: int = 512
    clip_width: int = 768              # input dim of the ClipCap MLP (hard-coded 768 at MFULL:1136; 1024 for ViT-L/14)

    @property
    def head_dim(self):
        return self.d_model // self.encoder_attention_heads

    @property
    def prompt_len(self):
        """visual-prompt tokens entering the encoder: prompt_size (ClipCap MLP) or map_size[-1] (token-mixing MLP)."""
        return self.prompt_size if self.prompt_mlp_type == "clipcap" else int(self.map_size[-1])

    def validate(self):
        if self.d_model not in (768, 1024):
            raise ValueError("d_model must be 768 or 1024 (prompt is reshaped to [B,P,768], MFULL:1143-1144,1275-1278)")
        if self.d_model % self.encoder_attention_heads or self.d_model // self.encoder_attention_heads != 64:
            raise ValueError("head_dim must be 64 (bart-base 768/12, bart-large 1024/16)")
        if self.prompt_mlp_type not in ("clipcap", "mlp"):
            raise ValueError(f"prompt_mlp_type must be 'clipcap' or 'mlp', got {self.prompt_mlp_type!r}")
        if self.prompt_mlp_type == "mlp":
            ms = self.map_size
            if not ms or len(ms) < 2 or any(int(m) <= 0 for m in ms):
                raise ValueError("--prompt_mlp_type mlp needs --map_size n_tokens h1 ... prompt_len (MFULL:76-108,1138)")
            if any(int(m) % 8 for m in ms[1:]):
                raise ValueError("map_size[1:] must be multiples of 8 (16-byte rows for the GEMMs); map_size[0] is free")
            if self.clip_width != 768:
                raise ValueError("prompt_mlp_type mlp keeps the ViT width as the prompt width: it needs a 768-wide CLIP tower "
                                 "(MFULL:1143,1277 feed it to Linear(768, 1024))")
        if not self.only_image and self.dim_common != self.d_model:
            raise ValueError("dim_common must equal d_model: face states are concatenated with name states (MFULL:668)")
        return self


@dataclass
class ClipVisionConfig:
    width: int = 1024
    layers: int = 24
    patch_size: int = 14
    image_size: int = 224
    output_dim: int = 768

    @property
    def heads(self):
        return self.width // 64

    @property
    def grid(self):
        return self.image_size // self.patch_size

    @property
    def tokens(self):
        return self.grid * self.grid + 1


# Dropout probabilities that live in the hub checkpoints' config.json: the reference builds its model with
# `BartForMultiModalGeneration.from_pretrained(plm_type)` (TRAIN:743), so it inherits THESE, not BartConfig's class defaults
# (attention_dropout = activation_dropout = 0.0, which only bart-large-cnn / -xsum publish).  The files are unreadable offline
# (SURVEY §8c); these are the published values of facebook/bart-base and facebook/bart-large (the fp32 re-upload copies
# bart-large's).  NB the kernels quantise p to k/256 (0.1 -> 26/256 = 0.1016; DESIGN §2 "Dropout").
HUB_MODEL_DROPOUTS = {
    "facebook/bart-base": dict(dropout=0.1, attention_dropout=0.1, activation_dropout=0.1),
    "facebook/bart-large": dict(dropout=0.1, attention_dropout=0.1, activation_dropout=0.1),
    "patrickvonplaten/bart-large-fp32": dict(dropout=0.1, attention_dropout=0.1, activation_dropout=0.1),
}


def bart_large_vit_l14(**kw):
    """BASELINE.json configs[1..2]: BART-large + CLIP ViT-L/14, full VACNIC, with the hub checkpoint's dropouts (keywords override)."""
    kw = dict(HUB_MODEL_DROPOUTS["facebook/bart-large"], **kw)
    return VacnicConfig(clip_width=1024, **kw).validate(), ClipVisionConfig()


def bart_base_vit_b32(**kw):
    """BASELINE.json configs[0]: BART-base + ViT-B/32 --only_image."""
    kw = dict(HUB_MODEL_DROPOUTS["facebook/bart-base"], **kw)
    c = VacnicConfig(d_model=768, encoder_layers=6, decoder_layers=6, encoder_attention_heads=12,
                     decoder_attention_heads=12, encoder_ffn_dim=3072, decoder_ffn_dim=3072,
                     enc_fusion_layer=list(range(6)), dim_common=768, only_image=True, clip_width=768, **kw).validate()
    return c, ClipVisionConfig(width=768, layers=12, patch_size=32, output_dim=512)

# Generation defaults that live in the hub checkpoints' config.json (NOT in BartConfig's own defaults): a model built by
# `from_pretrained(plm_type)` (TRAIN:743) carries them, and transformers 4.18 `generate()` falls back to them for every argument
# the caller leaves out — the reference's trainers pass only num_beams / max_length (TRAIN:513-520), its stand-alone generator also
# length_penalty (DDPINF:38,867).  The files are unreadable offline; these are the published values of
# facebook/bart-base and facebook/bart-large (the fp32 re-upload copies bart-large's).
HUB_GENERATION_DEFAULTS = {
    "facebook/bart-base": dict(no_repeat_ngram_size=3, early_stopping=True, forced_bos_token_id=0, forced_eos_token_id=2),
    "facebook/bart-large": dict(no_repeat_ngram_size=3, early_stopping=True, forced_bos_token_id=0, forced_eos_token_id=2),
    "patrickvonplaten/bart-large-fp32": dict(no_repeat_ngram_size=3, early_stopping=True, forced_bos_token_id=0, forced_eos_token_id=2),
}


def generation_defaults(plm_type=None):
    """kwargs for generate() that reproduce the reference's `model.generate(num_beams, max_length)` on a hub checkpoint.
    plm_type None = the shipped scripts' facebook/bart-large.  An UNKNOWN checkpoint name raises: its config.json may carry other
    values (bart-large-cnn: min_length 56, length_penalty 2.0, max_length 142 ...), and decoding it with bart-large's would
    silently differ from what the reference does — pass the checkpoint's values as explicit generate() keywords instead."""
    key = plm_type or "facebook/bart-large"
    if key not in HUB_GENERATION_DEFAULTS:
        raise KeyError(f"no recorded generation defaults for checkpoint {plm_type!r} (known: {sorted(HUB_GENERATION_DEFAULTS)}); "
                       "pass no_repeat_ngram_size / early_stopping / forced_bos_token_id / ... explicitly")
    return dict(HUB_GENERATION_DEFAULTS[key])
