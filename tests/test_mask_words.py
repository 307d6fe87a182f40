"""Host-side packing of explicit imagine masks into the kernels' 64-bit key words (refil_amd.modules.mixers.flex_qmix.
pack_mask_words, the path FlexQMixer.forward takes for the reference's imagine_groups=(Wmask, Imask) tensors,
flex_qmix.py:85-94) against the fixture's masks and the oracle's closed form."""
import torch

from golden_util import load
from oracle import refil_oracle as orc


def _bit(words, j):
    return ((words >> j) & 1).bool()


def test_pack_mask_words_matches_reference_masks():
    from refil_amd.modules.mixers.flex_qmix import pack_mask_words
    g = load("refil_abs_masked")
    z, cfg = g["z"], g["cfg"]
    em = g["batch"]["entity_mask"][:, :-1]                       # [B,T,ne]
    B, T, ne = em.shape
    na = cfg.n_agents
    Wm = torch.from_numpy(z["Wmask_noobs"])[:, None].repeat(1, T, 1, 1)
    Im = torch.from_numpy(z["Imask_noobs"])[:, None].repeat(1, T, 1, 1)
    mw, rb = pack_mask_words(Wm, Im, em, na)
    na_pad = 16 * ((na + 15) // 16)
    assert mw.shape == (B * T, 3, na_pad) and rb.shape == (B * T, 3) and mw.dtype == torch.int64
    mw = mw.reshape(B, T, 3, na_pad)
    emb = em.bool()
    for j in range(ne):
        # variant 0: the hypernets' default mask 1 - active_i active_j (flex_qmix.py:43-46)
        assert torch.equal(_bit(mw[:, :, 0, :na], j), emb[:, :, :na] | emb[:, :, j:j + 1])
        assert torch.equal(_bit(mw[:, :, 1, :na], j), Wm[:, :, :na, j].bool())
        assert torch.equal(_bit(mw[:, :, 2, :na], j), Im[:, :, :na, j].bool())
        assert torch.equal(_bit(rb.reshape(B, T, 3)[:, :, 2], j), emb[:, :, j])
    for j in range(ne, 64):                                       # padded keys are masked, padded agents all ones
        assert _bit(mw[:, :, :, :na], j).all()
    assert (mw[:, :, :, na:] == -1).all()
    assert (rb.reshape(B, T, 3)[:, :, :2] == 0).all()
    # the oracle's closed form of the imagine masks gives the same words from the partition bits
    W2, I2 = orc.imagine_masks(g["bits"], g["batch"]["entity_mask"][:, 0])
    mw2, _ = pack_mask_words(W2[:, None].expand(B, T, ne, ne), I2[:, None].expand(B, T, ne, ne), em, na)
    assert torch.equal(mw2.reshape(B, T, 3, na_pad), mw)
