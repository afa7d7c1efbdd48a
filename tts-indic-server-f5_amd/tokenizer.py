"""Character tokenizer of the reference (F/model/utils.py:88-131): vocab file -> {char: idx}, unknown -> 0,
ragged batch padded with -1."""
from __future__ import annotations

import torch


def get_tokenizer(vocab_path: str, tokenizer: str = "custom"):
    """get_tokenizer(dataset_name, "custom") (F/model/utils.py:123-129): one token per line, idx = line number."""
    if tokenizer != "custom":
        raise ValueError("only the 'custom' tokenizer (explicit vocab.txt path) is on the inference path")
    vocab_char_map = {}
    with open(vocab_path, "r", encoding="utf-8") as f:
        for i, line in enumerate(f):
            vocab_char_map[line[:-1]] = i
    return vocab_char_map, len(vocab_char_map)


def list_str_to_idx(text, vocab_char_map, padding_value: int = -1) -> torch.Tensor:
    """F/model/utils.py:88-95."""
    rows = [torch.tensor([vocab_char_map.get(c, 0) for c in t], dtype=torch.long) for t in text]
    return torch.nn.utils.rnn.pad_sequence(rows, padding_value=padding_value, batch_first=True)
