"""Host side of the text front-end: tokenizer-vocabulary hookup for the on-device word-piece
aggregation, and the `sents` lists the reference hands around.

Follows text_encoder.py:32-90 (aggregate_tokens): a caption's tokens are read up to and including the first
[SEP]; a token that does not start with '##' opens a new word, '##' pieces are appended (without the '##') to the open
word; [SEP] closes the open word and is a word of its own; the list is padded with '[PAD]' to the token count.  Without a
[SEP] the last open word is never closed (dropped).  cap_lens = number of words not starting with '[' plus one
(medmoe_module.py:221-223).

The embeddings themselves are aggregated on the device (medmoe_text_aggregate with the segment map of
VocabTables.segment_map); strings stay on the host because that is where the reference's consumers read them.
"""
from typing import Dict, List, Sequence, Union

import torch

from .engine import VocabTables

Vocabulary = Union[Dict[int, str], Sequence[str]]


def _as_list(idxtoword: Vocabulary) -> List[str]:
    if isinstance(idxtoword, dict):
        n = max(idxtoword) + 1
        return [idxtoword.get(i, "[UNK]") for i in range(n)]
    return list(idxtoword)


def vocab_tables(idxtoword: Vocabulary, device, sep_token: str = "[SEP]") -> VocabTables:
    """Device tables for VocabTables.segment_map from a tokenizer vocabulary (text_encoder.py:23: idxtoword)."""
    words = _as_list(idxtoword)
    cont = torch.tensor([w.startswith("##") for w in words], dtype=torch.bool, device=device)
    br = torch.tensor([w.startswith("[") for w in words], dtype=torch.bool, device=device)
    vt = VocabTables(cont, br)
    vt.sep_id = words.index(sep_token)
    return vt


def merge_sents(ids, idxtoword: Vocabulary) -> List[List[str]]:
    """The `sents` output of BertEncoder.forward (text_encoder.py:45-76,130-144): merged words per caption."""
    words = _as_list(idxtoword)
    rows = ids.tolist() if hasattr(ids, "tolist") else [list(r) for r in ids]
    out = []
    for row in rows:
        sent: List[str] = []
        bank: List[str] = []
        for tid in row:
            w = words[int(tid)]
            if w == "[SEP]":
                sent.append("".join(bank))
                sent.append(w)
                bank = []
                break
            if w.startswith("##"):
                bank.append(w[2:])
            elif not bank:
                bank.append(w)
            else:
                sent.append("".join(bank))
                bank = [w]
        sent += ["[PAD]"] * (len(row) - len(sent))
        out.append(sent)
    return out


def cap_lens_from_sents(sents: List[List[str]]) -> List[int]:
    """medmoe_module.py:221-223."""
    return [len([w for w in s if not w.startswith("[")]) + 1 for s in sents]
