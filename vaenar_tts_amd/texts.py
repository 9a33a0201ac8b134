"""Text front-end in front of the text->mel path: counterpart of reference texts/texts.py:1-142 (the keithito/tacotron English
cleaners) and of `text_to_array` (datasets/datasets.py:253-262 LJSpeech, :365-378 DataBaker).

The reference leans on two third-party packages that are not dependencies here (environment.yml: inflect, unidecode; pypinyin for
Mandarin):

* ``inflect.engine().number_to_words`` -- restated below for exactly the call shapes texts.py uses (:106-123): cardinals with
  ``andword=''`` (comma-separated thousands groups, hyphenated tens-units), the year reading ``group=2, zero='oh'`` and ordinals
  of digit strings ("21st" -> "twenty-first");
* ``unidecode`` -- ASCII transliteration.  Here: NFKD decomposition with combining marks dropped plus a small table for the
  punctuation and Latin letters NFKD leaves alone; characters outside that are dropped (unidecode would spell most of them out).
  ASCII text -- the LJSpeech transcripts -- passes through unchanged, as with unidecode;
* ``pypinyin`` (DataBaker ``text_to_array``) -- not restated: `pinyin_to_array` takes the TONE3 syllables
  ("ni3 hao3") the reference obtains from ``pinyin(text, style=Style.TONE3, neutral_tone_with_five=True)``.
"""
import re
import unicodedata

_whitespace_re = re.compile(r"\s+")

# texts.py:9-28 -- (abbreviation, expansion), matched case-insensitively as a whole word followed by a period
_ABBREVIATIONS = [
    ("mrs", "misess"), ("mr", "mister"), ("dr", "doctor"), ("st", "saint"), ("co", "company"), ("jr", "junior"), ("maj", "major"),
    ("gen", "general"), ("drs", "doctors"), ("rev", "reverend"), ("lt", "lieutenant"), ("hon", "honorable"), ("sgt", "sergeant"),
    ("capt", "captain"), ("esq", "esquire"), ("ltd", "limited"), ("col", "colonel"), ("ft", "fort"),
]
_abbreviations = [(re.compile(r"\b%s\." % a, re.IGNORECASE), b) for a, b in _ABBREVIATIONS]

# ---- number_to_words (inflect) for the shapes texts.py needs --------------------------------------------------------------
_UNITS = ["zero", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten", "eleven", "twelve", "thirteen",
          "fourteen", "fifteen", "sixteen", "seventeen", "eighteen", "nineteen"]
_TENS = ["", "", "twenty", "thirty", "forty", "fifty", "sixty", "seventy", "eighty", "ninety"]
_SCALES = ["", " thousand", " million", " billion", " trillion", " quadrillion", " quintillion", " sextillion"]
_ORDINAL_WORD = {"one": "first", "two": "second", "three": "third", "five": "fifth", "eight": "eighth", "nine": "ninth",
                 "twelve": "twelfth"}


def _below_100(n, zero="zero"):
    if n < 20:
        return zero if n == 0 else _UNITS[n]
    return _TENS[n // 10] + ("-" + _UNITS[n % 10] if n % 10 else "")


def _below_1000(n, andword):
    h, r = divmod(n, 100)
    if not h:
        return _below_100(r)
    if not r:
        return _UNITS[h] + " hundred"
    return _UNITS[h] + " hundred " + (andword + " " if andword else "") + _below_100(r)


def number_to_words(num, andword="and", zero="zero", group=0):
    """inflect.engine().number_to_words(num, andword=, zero=, group=) for non-negative integers (group 0 or 2)."""
    n = int(num)
    if group == 2:                                       # digit pairs from the left: 1984 -> "nineteen, eighty-four"
        s = str(n)
        out = []
        for i in range(0, len(s), 2):
            pair = s[i:i + 2]
            if len(pair) == 1:
                out.append(zero if pair == "0" else _UNITS[int(pair)])
            elif pair[0] == "0":
                out.append("%s %s" % (zero, zero if pair[1] == "0" else _UNITS[int(pair[1])]))
            else:
                out.append(_below_100(int(pair)))
        return ", ".join(out)
    if n == 0:
        return zero
    groups = []
    while n:
        n, r = divmod(n, 1000)
        groups.append(r)
    parts = []
    for i in range(len(groups) - 1, -1, -1):
        if groups[i]:
            # inflect joins the last group with "and" when it is below 100 and something precedes it: "one thousand and five"
            if i == 0 and groups[0] < 100 and parts and andword:
                parts[-1] = parts[-1] + " " + andword + " " + _below_100(groups[0])
                continue
            parts.append(_below_1000(groups[i], andword) + _SCALES[i])
    return ", ".join(parts)


def ordinal_words(num):
    """inflect's number_to_words on an ordinal digit string ("21st", "100th"): cardinal words with the last word made ordinal."""
    words = number_to_words(int(re.match(r"\d+", str(num)).group(0)))
    head, sep, last = words.rpartition(" ")
    pre, hy, unit = last.rpartition("-")
    if unit in _ORDINAL_WORD:
        unit = _ORDINAL_WORD[unit]
    elif unit.endswith("y"):
        unit = unit[:-1] + "ieth"
    else:
        unit = unit + "th"
    return head + sep + pre + hy + unit


_comma_number_re = re.compile(r"([0-9][0-9\,]+[0-9])")
_decimal_number_re = re.compile(r"([0-9]+\.[0-9]+)")
_pounds_re = re.compile(r"£([0-9\,]*[0-9]+)")
_dollars_re = re.compile(r"\$([0-9\.\,]*[0-9]+)")
_ordinal_re = re.compile(r"[0-9]+(st|nd|rd|th)")
_number_re = re.compile(r"[0-9]+")


def _expand_dollars(m):                                    # texts.py:84-103
    match = m.group(1)
    parts = match.split(".")
    if len(parts) > 2:
        return match + " dollars"
    dollars = int(parts[0]) if parts[0] else 0
    cents = int(parts[1]) if len(parts) > 1 and parts[1] else 0
    if dollars and cents:
        return "%s %s, %s %s" % (dollars, "dollar" if dollars == 1 else "dollars", cents, "cent" if cents == 1 else "cents")
    if dollars:
        return "%s %s" % (dollars, "dollar" if dollars == 1 else "dollars")
    if cents:
        return "%s %s" % (cents, "cent" if cents == 1 else "cents")
    return "zero dollars"


def _expand_number(m):                                     # texts.py:110-123
    num = int(m.group(0))
    if 1000 < num < 3000:
        if num == 2000:
            return "two thousand"
        if 2000 < num < 2010:
            return "two thousand " + number_to_words(num % 100)
        if num % 100 == 0:
            return number_to_words(num // 100) + " hundred"
        return number_to_words(num, andword="", zero="oh", group=2).replace(", ", " ")
    return number_to_words(num, andword="")


def normalize_numbers(text):                               # texts.py:126-133
    text = re.sub(_comma_number_re, lambda m: m.group(1).replace(",", ""), text)
    text = re.sub(_pounds_re, r"\1 pounds", text)
    text = re.sub(_dollars_re, _expand_dollars, text)
    text = re.sub(_decimal_number_re, lambda m: m.group(1).replace(".", " point "), text)
    text = re.sub(_ordinal_re, lambda m: ordinal_words(m.group(0)), text)
    text = re.sub(_number_re, _expand_number, text)
    return text


_ASCII_EXTRA = {"‘": "'", "’": "'", "“": '"', "”": '"', "–": "-", "—": "--", "…": "...",
                "ß": "ss", "æ": "ae", "Æ": "AE", "ø": "o", "Ø": "O", "œ": "oe", "Œ": "OE",
                "đ": "d", "Đ": "D", "ł": "l", "Ł": "L", "þ": "th", "Þ": "Th", "ð": "d", "Ð": "D",
                " ": " ", "«": "<<", "»": ">>"}


def convert_to_ascii(text):                                # texts.py:48-49 (unidecode; see the module docstring)
    out = []
    for ch in text:
        if ord(ch) < 128:
            out.append(ch)
        elif ch in _ASCII_EXTRA:
            out.append(_ASCII_EXTRA[ch])
        else:
            out.append("".join(c for c in unicodedata.normalize("NFKD", ch) if ord(c) < 128))
    return "".join(out)


def expand_abbreviations(text):                            # texts.py:31-34
    for regex, replacement in _abbreviations:
        text = re.sub(regex, replacement, text)
    return text


def expand_numbers(text):
    return normalize_numbers(text)


def lowercase(text):
    return text.lower()


def collapse_whitespace(text):
    return re.sub(_whitespace_re, " ", text)


def basic_cleaners(text):                                  # texts.py:52-56
    return collapse_whitespace(lowercase(text))


def transliteration_cleaners(text):                        # texts.py:59-64
    return collapse_whitespace(lowercase(convert_to_ascii(text)))


def english_cleaners(text):                                # texts.py:67-74
    text = convert_to_ascii(text)
    text = lowercase(text)
    text = expand_numbers(text)
    text = expand_abbreviations(text)
    return collapse_whitespace(text)


def text_to_array(text, hps):
    """LJSpeech.text_to_array (datasets/datasets.py:253-262): english_cleaners, bos/eos, character ids.  A character outside
    ``hps.Texts.characters`` raises KeyError, as in the reference."""
    t = hps.Texts
    symbol_to_id = {s: i for i, s in enumerate(t.characters)}
    return [symbol_to_id[s] for s in t.bos + english_cleaners(text) + t.eos]


def pinyin_to_array(syllables, hps):
    """DataBaker.text_to_array (datasets/datasets.py:365-378) after its pypinyin call: TONE3 syllables (list, or one string
    separated by blanks) -> lower-cased, blank-joined, bos/eos, character ids."""
    if isinstance(syllables, str):
        syllables = syllables.split()
    t = hps.Texts
    symbol_to_id = {s: i for i, s in enumerate(t.characters)}
    return [symbol_to_id[s] for s in t.bos + " ".join(p.lower() for p in syllables) + t.eos]


def pad_batch(arrays):
    """inference.py:51-53: (ids [B, max_len] int32 zero padded, lengths [B] int32)."""
    import numpy as np
    lens = np.array([len(a) for a in arrays], np.int32)
    ids = np.zeros((len(arrays), int(lens.max())), np.int32)
    for i, a in enumerate(arrays):
        ids[i, :len(a)] = a
    return ids, lens
