"""Synthetic bitnet-b1.58 weights and prompts (there is no model file in the build
environment; SURVEY.md 8d fixes the recipe).  Data only: no arithmetic of the path.

  * I2_S codes: uniform 2-bit codes from a counter-based PRNG, seed 42, one stream per
    tensor (crates/bitnet-models/tests/qk256_avx2_correctness.rs:72-88 uses uniform codes);
  * norm gammas ~ U(0.5, 1.5) / (w_rms * sqrt(hidden)), w_rms = the format's weight rms
    (1.58 for unscaled {-2,-1,+1,+2}, 0.181 for ternary x block scales), so that projections
    of a normalised vector stay O(1) and greedy tokens vary;
  * tied embedding table f16 ~ N(0, 1);
  * prompt ids (1000 + 37 i) mod vocab with BOS first.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict

import numpy as np

BITNET_2B_4T = dict(hidden=2560, n_layers=30, n_heads=20, n_kv_heads=5, head_dim=128, ffn=6912, vocab=128256, max_pos=4096, eps=1e-5, rope_theta=10000.0)
# crates/bitnet-models/src/qk256_utils.rs:95-104 (shape), T:957 (eps default), crates/bitnet-rope/src/lib.rs:11 (theta default)

PROJ = ("q", "k", "v", "o", "gate", "up", "down")


@dataclass
class ModelConfig:
    hidden: int = 2560
    n_layers: int = 30
    n_heads: int = 20
    n_kv_heads: int = 5
    head_dim: int = 128
    ffn: int = 6912
    vocab: int = 128256
    max_pos: int = 4096
    eps: float = 1e-5
    rope_theta: float = 10000.0

    def shapes(self):
        """[rows=out, cols=in] per projection (crates/bitnet-models/src/qk256_utils.rs:19-55)."""
        qd, kd = self.n_heads * self.head_dim, self.n_kv_heads * self.head_dim
        return {"q": (qd, self.hidden), "k": (kd, self.hidden), "v": (kd, self.hidden), "o": (self.hidden, qd),
                "gate": (self.ffn, self.hidden), "up": (self.ffn, self.hidden), "down": (self.hidden, self.ffn)}

    def asdict(self):
        return asdict(self)


def _stream(seed: int, *ids: int) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=seed, counter=[*ids, 0, 0, 0][:4]))


def qk256_codes(rows: int, cols: int, seed: int, layer: int, proj: int) -> np.ndarray:
    stride = -(-cols // 256) * 64
    return _stream(seed, layer, proj).integers(0, 256, rows * stride, dtype=np.uint8)


def ternary_weights(rows: int, cols: int, block: int, seed: int, layer: int, proj: int):
    """Codes from {0,1,3} with P(0)=.5, P(+1)=P(-1)=.25 and scales 1/((i%100)+1)
    (crates/bitnet-quantization/benches/qk256_gemv.rs:41-43), scaled to keep O(1) outputs."""
    rng = _stream(seed, layer, proj, 1)
    assert cols % 4 == 0
    # per 2-bit field: low bit ~ Bernoulli(.5) (code is odd: +-1), high bit = low & Bernoulli(.5)
    # (code 3 = -1)  ->  P(0)=.5, P(1)=.25, P(3)=.25, never 2
    a = rng.integers(0, 256, rows * cols // 4, dtype=np.uint8)
    b = rng.integers(0, 256, rows * cols // 4, dtype=np.uint8)
    packed = (a & 0x55) | (((a & b) & 0x55) << 1)
    nblk = -(-cols // block)
    # BitNet32-F16 stores one f16 scale per 32 weights: the synthetic scales are f16 values (held as f32)
    scales = (2.0 / ((np.arange(rows * nblk) % 100) + 1)).astype(np.float16).astype(np.float32)
    return packed.reshape(-1), scales


# rms of one weight: QK256 codes uniform over {-2,-1,+1,+2} -> sqrt(2.5) = 1.58; ternary {0,+-1} with P(+-1) = .5 times the
# scales 2/((i%100)+1) -> sqrt(.5 * mean(s^2)) = sqrt(.5 * 4 * 0.01635) = 0.181
W_RMS = {"qk256": 1.58, "i2s": 0.181}


def make_layer(cfg: ModelConfig, layer: int, seed: int = 42, fmt: str = "qk256", block: int = 32) -> dict:
    g = _stream(seed, layer, 100)
    # gammas sized for the format's weight rms, so that the projections of a normalised vector are O(1) in BOTH formats
    # (round 1 used the QK256 figure for the ternary format too: projections ~0.1, the residual stream stayed the token's own
    # embedding and the tied logits returned the input token forever)
    norm_scale = 1.0 / (W_RMS["qk256" if fmt == "qk256" else "i2s"] * np.sqrt(cfg.hidden))
    out = {
        "attn_norm": (g.uniform(0.5, 1.5, cfg.hidden) * norm_scale).astype(np.float32),
        "ffn_norm": (g.uniform(0.5, 1.5, cfg.hidden) * norm_scale).astype(np.float32),
    }
    for i, name in enumerate(PROJ):
        rows, cols = cfg.shapes()[name]
        if fmt == "qk256":
            out[name] = qk256_codes(rows, cols, seed, layer, i)
        else:
            out[name], out[name + "_scales"] = ternary_weights(rows, cols, block, seed, layer, i)
    return out


def make_globals(cfg: ModelConfig, seed: int = 42) -> dict:
    g = _stream(seed, 1 << 20)
    emb = g.standard_normal((cfg.vocab, cfg.hidden), dtype=np.float32).astype(np.float16)
    return {"embed_f16": emb.view(np.uint16).reshape(-1), "final_norm": g.uniform(0.5, 1.5, cfg.hidden).astype(np.float32)}


def prompt(n: int, vocab: int, bos: int = 128000) -> np.ndarray:
    ids = (1000 + 37 * np.arange(n)) % vocab
    ids[0] = bos % vocab
    return ids.astype(np.int32)
