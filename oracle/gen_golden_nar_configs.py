"""TEST INFRASTRUCTURE ONLY.  Shapes / settings shared by oracle/gen_golden_nar.py and the tests of the NAR decoder."""
import torch

import nar_oracle as N
from gen_golden_configs import seeded

CFG = N.NarConfig(embed_dim=64, ffn_dim=128, layers=2, heads=4, vocab=1004)
SETTINGS = [dict(max_iter=4, beam_size=1, adaptive=True), dict(max_iter=9, beam_size=1, adaptive=False, retain_history=True),
            dict(max_iter=3, beam_size=3, adaptive=True), dict(max_iter=0, beam_size=1, adaptive=True)]


class Dict1004:
    def bos(self): return 0
    def pad(self): return 1
    def eos(self): return 2
    def unk(self): return 3
    def __len__(self): return CFG.vocab


def encoder_out(B, S, lens, seed):
    x = seeded((S, B, CFG.embed_dim), seed)
    pad = torch.arange(S)[None, :] >= lens[:, None]
    return {"encoder_out": [x], "encoder_padding_mask": [pad], "encoder_embedding": [], "encoder_states": [], "src_tokens": [], "src_lengths": []}


# the speech encoder at fixture size (embed / heads match CFG so its output feeds the fixture decoder)
ENC_CFG = N.NarEncoderConfig(input_dim=80, conv_channels=128, kernel_sizes=(5, 5), embed_dim=64, ffn_dim=128, layers=2, heads=4)
