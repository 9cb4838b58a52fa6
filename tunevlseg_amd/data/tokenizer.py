"""Byte-level BPE tokenizer of CLIP (the ``AutoTokenizer.from_pretrained("CIDAS/clipseg-rd64")`` of the reference's data pipeline,
``configs/experiment/coop/clipseg.yaml:130-132``; same algorithm and vocabulary as the reference's vendored
``src/models/components/denseclip/untils.py:69-175``).

The merge table is NOT shipped with this repo: pass ``bpe_path`` (OpenAI's ``bpe_simple_vocab_16e6.txt.gz`` or a HF tokenizer
directory / ``merges.txt``), or set ``TVL_CLIP_BPE``.  Calling the object mirrors the HF call the dataset makes
(``image_text_mask_dataset.py:91``): ``tokenizer(prompt) -> {"input_ids": [BOS, ..., EOS], "attention_mask": [1, ...]}``.

Algorithm (published with CLIP): lower-case, collapse whitespace, split with the CLIP regex, map every UTF-8 byte of a piece to a
printable code point, then greedily apply the lowest-ranked merge until none applies; ids = 256 byte symbols, their 256
end-of-word forms, the merges in rank order, then the two specials.
"""
from __future__ import annotations

import gzip
import html
import os
from pathlib import Path

import regex

N_MERGES = 49152 - 256 - 2  # CLIP's vocabulary: 49 408 = 512 byte symbols + 48 894 merges + 2 specials
_SPLIT = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)


def _byte_symbols() -> list[str]:
    """Printable stand-in for each byte value: the visible Latin-1 bytes map to themselves, the other 68 to U+0100 onward."""
    visible = [*range(0x21, 0x7F), *range(0xA1, 0xAD), *range(0xAE, 0x100)]
    table, spare = {}, 0x100
    for b in visible:
        table[b] = chr(b)
    for b in range(256):
        if b not in table:
            table[b] = chr(spare)
            spare += 1
    return [table[b] for b in range(256)]


def _read_merges(path: Path) -> list[tuple[str, str]]:
    if path.is_dir():
        path = path / "merges.txt"
    opener = gzip.open if path.suffix == ".gz" else open
    with opener(path, "rt", encoding="utf-8") as fh:
        lines = fh.read().split("\n")
    pairs = [tuple(ln.split()) for ln in lines[1:N_MERGES + 1]]  # line 0 is a version header
    if len(pairs) != N_MERGES or any(len(p) != 2 for p in pairs):
        raise ValueError(f"{path}: expected {N_MERGES} merge rules after the header line")
    return pairs  # type: ignore[return-value]


class ClipBpeTokenizer:
    bos_token, eos_token = "<|startoftext|>", "<|endoftext|>"

    def __init__(self, bpe_path: str | os.PathLike | None = None, pad_token_id: int | None = None, model_max_length: int = 77):
        bpe_path = bpe_path or os.environ.get("TVL_CLIP_BPE")
        if not bpe_path or not Path(bpe_path).exists():
            raise FileNotFoundError("CLIP BPE merge table not found: pass bpe_path= (bpe_simple_vocab_16e6.txt.gz, a HF tokenizer "
                                    "directory or its merges.txt) or set TVL_CLIP_BPE")
        self.bpe_path = Path(bpe_path)
        merges = _read_merges(Path(bpe_path))
        self._rank = {pair: i for i, pair in enumerate(merges)}
        symbols = _byte_symbols()
        self._byte_symbol = symbols
        vocab = [*sorted_byte_vocab(symbols), *(s + "</w>" for s in sorted_byte_vocab(symbols)), *("".join(m) for m in merges),
                 self.bos_token, self.eos_token]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = dict(enumerate(vocab))
        self.bos_token_id, self.eos_token_id = self.encoder[self.bos_token], self.encoder[self.eos_token]
        self.pad_token_id = self.eos_token_id if pad_token_id is None else int(pad_token_id)  # CLIP pads with <|endoftext|>
        self.model_max_length = model_max_length
        self._memo: dict[str, tuple[str, ...]] = {}

    def __len__(self) -> int:
        return len(self.encoder)

    # ------------------------------------------------------------------ BPE of one pre-token
    def _merge_word(self, piece: str) -> tuple[str, ...]:
        hit = self._memo.get(piece)
        if hit is not None:
            return hit
        parts = [*piece[:-1], piece[-1] + "</w>"]
        while len(parts) > 1:
            best, where = None, -1
            for i in range(len(parts) - 1):
                r = self._rank.get((parts[i], parts[i + 1]))
                if r is not None and (best is None or r < best):
                    best, where = r, i
            if best is None:
                break
            a, b = parts[where], parts[where + 1]
            merged, i = [], 0
            while i < len(parts):  # every non-overlapping occurrence of the pair, left to right
                if i + 1 < len(parts) and parts[i] == a and parts[i + 1] == b:
                    merged.append(a + b)
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        out = tuple(parts)
        self._memo[piece] = out
        return out

    def encode(self, text: str) -> list[int]:
        """BPE ids of ``text`` without the start / end specials."""
        text = regex.sub(r"\s+", " ", html.unescape(text).strip()).strip().lower()
        ids: list[int] = []
        for piece in _SPLIT.findall(text):
            if piece in (self.bos_token, self.eos_token):
                ids.append(self.encoder[piece])
                continue
            mapped = "".join(self._byte_symbol[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[tok] for tok in self._merge_word(mapped))
        return ids

    def decode(self, ids) -> str:
        text = "".join(self.decoder[int(i)] for i in ids)
        inverse = {s: b for b, s in enumerate(self._byte_symbol)}
        raw = bytearray()
        for ch in text.replace("</w>", " "):
            raw.append(inverse[ch]) if ch in inverse else raw.extend(ch.encode("utf-8"))
        return raw.decode("utf-8", errors="replace")

    def __call__(self, text, add_special_tokens: bool = True, truncation: bool = False, max_length: int | None = None,
                 return_tensors: str | None = None, return_attention_mask: bool = True, **_ignored):
        """HF-tokenizer style call on one string or a list of strings (no padding here: the collator pads to the longest)."""
        single = isinstance(text, str)
        rows = []
        for t in ([text] if single else list(text)):
            ids = self.encode(t)
            if add_special_tokens:
                ids = [self.bos_token_id, *ids, self.eos_token_id]
            if truncation:
                limit = max_length or self.model_max_length
                if len(ids) > limit:
                    ids = [*ids[:limit - 1], self.eos_token_id] if add_special_tokens else ids[:limit]
            rows.append(ids)
        if return_tensors == "pt":
            import torch

            if len({len(r) for r in rows}) != 1:
                raise ValueError("return_tensors='pt' needs rows of equal length (pad with PadToLongestCollator)")
            out = {"input_ids": torch.tensor(rows, dtype=torch.long)}
            if return_attention_mask:
                out["attention_mask"] = torch.ones_like(out["input_ids"])
            return _AttrDict(out)
        out = {"input_ids": rows[0] if single else rows}
        if return_attention_mask:
            out["attention_mask"] = [1] * len(rows[0]) if single else [[1] * len(r) for r in rows]
        return _AttrDict(out)


def sorted_byte_vocab(symbols: list[str]) -> list[str]:
    """The 256 byte symbols in CLIP's vocabulary order: visible bytes first (in byte order), then the re-mapped ones."""
    visible = [*range(0x21, 0x7F), *range(0xA1, 0xAD), *range(0xAE, 0x100)]
    rest = [b for b in range(256) if b not in set(visible)]
    return [symbols[b] for b in (*visible, *rest)]


class _AttrDict(dict):
    """dict that also answers ``.input_ids`` (the learners read the attribute, ``coop_context_learner.py:71-77``)."""

    __getattr__ = dict.__getitem__
