"""Text front-end for ``Synthesizer.tts``: cleaners -> characters -> token ids.

Mirror of the reference's ``TextProcessor.tokenize`` / ``tokens_to_indices`` (core/processors.py:32-52) and
``base_cleaners`` (utils/text/cleaners.py:58-67, utils/text/numbers.py:62-69).  Host string processing; the
reference spells numbers with the third-party ``inflect`` package, which is not a dependency here: the small
English number speller below covers cardinals, ordinals, years, decimals and currency the way the reference's
regular-expression pipeline feeds them.  Grapheme-to-phoneme conversion (``use_g2p``, g2p_en) is not provided.
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional

from .configs import TextConfig

_ONES = ["zero", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten", "eleven", "twelve", "thirteen",
         "fourteen", "fifteen", "sixteen", "seventeen", "eighteen", "nineteen"]
_TENS = ["", "", "twenty", "thirty", "forty", "fifty", "sixty", "seventy", "eighty", "ninety"]
_SCALES = [(10 ** 9, "billion"), (10 ** 6, "million"), (1000, "thousand")]
_ORD = {"one": "first", "two": "second", "three": "third", "five": "fifth", "eight": "eighth", "nine": "ninth", "twelve": "twelfth"}


def _below_1000(n: int) -> str:
    words = []
    if n >= 100:
        words += [_ONES[n // 100], "hundred"]
        n %= 100
    if n >= 20:
        words.append(_TENS[n // 10] + ("-" + _ONES[n % 10] if n % 10 else ""))
    elif n > 0 or not words:
        words.append(_ONES[n])
    return " ".join(words)


def number_to_words(n: int) -> str:
    if n < 0:
        return "minus " + number_to_words(-n)
    if n < 1000:
        return _below_1000(n)
    parts = []
    for scale, name in _SCALES:
        if n >= scale:
            parts.append(number_to_words(n // scale) + " " + name)
            n %= scale
    if n:
        parts.append(_below_1000(n))
    return ", ".join(parts)


def _with_and(n: int) -> str:
    """Cardinal words with inflect's default andword: "and" before the final below-100 part when something precedes it
    ("one hundred and one", "one thousand and five", "one thousand, two hundred and thirty-four")."""
    if n < 100:
        return _below_1000(n)
    groups = []
    rest = n
    for scale, name in _SCALES:
        if rest >= scale:
            groups.append(_with_and(rest // scale) + " " + name)
            rest %= scale
    if not rest:
        return ", ".join(groups)
    h, lo = divmod(rest, 100)
    if h:
        groups.append(_ONES[h] + " hundred" + (" and " + _below_1000(lo) if lo else ""))
        return ", ".join(groups)
    return ", ".join(groups) + " and " + _below_1000(lo)


def ordinal_to_words(n: int) -> str:
    """The reference spells ordinals with inflect's defaults (utils/text/numbers.py:42-43: number_to_words(m.group(0)),
    andword="and"), unlike cardinals (andword="")."""
    words = _with_and(n)
    head, sep, last = words.rpartition(" ") if " " in words and "-" not in words.rsplit(" ", 1)[-1] else ("", "", words)
    stem, dash, unit = last.rpartition("-")
    if unit in _ORD:
        unit = _ORD[unit]
    elif unit.endswith("y"):
        unit = unit[:-1] + "ieth"
    else:
        unit = unit + "th"
    return (head + sep if head else "") + (stem + dash if stem else "") + unit


def _spell_year_or_number(m: "re.Match") -> str:
    n = int(m.group(0))
    if 1000 < n < 3000:
        if n == 2000:
            return "two thousand"
        if 2000 < n < 2010:
            return "two thousand " + number_to_words(n % 100)
        if n % 100 == 0:
            return number_to_words(n // 100) + " hundred"
        hi, lo = divmod(n, 100)
        return number_to_words(hi) + " " + ("oh " + _ONES[lo] if lo < 10 else number_to_words(lo))
    return number_to_words(n)


def _spell_dollars(m: "re.Match") -> str:
    parts = m.group(1).split(".")
    if len(parts) > 2:
        return m.group(1) + " dollars"
    dollars = int(parts[0]) if parts[0] else 0
    cents = int(parts[1]) if len(parts) > 1 and parts[1] else 0
    out = []
    if dollars:
        out.append(f"{dollars} dollar" + ("" if dollars == 1 else "s"))
    if cents:
        out.append(f"{cents} cent" + ("" if cents == 1 else "s"))
    return ", ".join(out) if out else "zero dollars"


def normalize_numbers(text: str) -> str:
    text = re.sub(r"([0-9][0-9\,]+[0-9])", lambda m: m.group(1).replace(",", ""), text)
    text = re.sub(r"£([0-9\,]*[0-9]+)", r"\1 pounds", text)
    text = re.sub(r"\$([0-9\.\,]*[0-9]+)", _spell_dollars, text)
    text = re.sub(r"([0-9]+\.[0-9]+)", lambda m: m.group(1).replace(".", " point "), text)
    text = re.sub(r"([0-9]+)(st|nd|rd|th)", lambda m: ordinal_to_words(int(m.group(1))), text)
    return re.sub(r"[0-9]+", _spell_year_or_number, text)


_ABBREVIATIONS = [(re.compile(r"\b%s\." % k, re.IGNORECASE), v) for k, v in [
    ("mrs", "misess"), ("mr", "mister"), ("dr", "doctor"), ("st", "saint"), ("co", "company"), ("jr", "junior"),
    ("maj", "major"), ("gen", "general"), ("drs", "doctors"), ("rev", "reverend"), ("lt", "lieutenant"),
    ("hon", "honorable"), ("sgt", "sergeant"), ("capt", "captain"), ("esq", "esquire"), ("ltd", "limited"),
    ("col", "colonel"), ("ft", "fort")]]
_INVALID = re.compile(r"[\[\]~`@#$%^&*()\-_+=|\"\'<>/]")


def base_cleaners(text: str, language: str = "english") -> str:
    text = text.lower()
    if language == "english":
        text = normalize_numbers(text)
        for rx, rep in _ABBREVIATIONS:
            text = rx.sub(rep, text)
    text = _INVALID.sub(" ", text)
    return re.sub(r"\s+", " ", text)


class TextProcessor:
    def __init__(self, config: TextConfig):
        self.config = config
        self.token_map: Optional[Dict[str, int]] = config.token_map
        self.all_unique_tokens = set()

    def tokenize(self, text: str) -> List[str]:
        for cleaner in (self.config.cleaners or []):
            if cleaner != "base_cleaners":
                raise KeyError(cleaner)
            text = base_cleaners(text, self.config.language)
        if self.config.use_g2p:
            raise NotImplementedError("grapheme-to-phoneme tokenisation (g2p_en) is not part of the MI355X forward path")
        tokens = list(text)
        self.all_unique_tokens.update(tokens)
        return tokens

    def generate_token_map(self) -> Dict[str, int]:
        self.token_map = {sym: i for i, sym in enumerate(sorted(self.all_unique_tokens))}
        self.config.token_map = self.token_map
        self.config.n_tokens = len(self.token_map)
        return self.token_map

    def tokens_to_indices(self, tokens: List[str]) -> List[int]:
        assert self.token_map is not None, "token_map not yet generated, use TextProcessor.generate_token_map() to generate it"
        return [self.token_map[t] for t in tokens]
