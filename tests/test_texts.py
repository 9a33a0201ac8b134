"""Text front-end (vaenar_tts_amd/texts.py, counterpart of reference texts/texts.py and datasets.py text_to_array).
inflect / unidecode are not installed (parity unpinned against them); the cases below are the documented behaviour of
inflect.number_to_words for the call shapes texts.py:106-123 uses and the well-known outputs of the keithito English cleaners."""
import numpy as np
import pytest

from vaenar_tts_amd import texts as T
from vaenar_tts_amd.configs import LJHPS, DataBakerHPS


@pytest.mark.parametrize("n,words", [
    (0, "zero"), (7, "seven"), (13, "thirteen"), (21, "twenty-one"), (40, "forty"), (100, "one hundred"), (101, "one hundred one"),
    (110, "one hundred ten"), (999, "nine hundred ninety-nine"), (1000, "one thousand"), (1005, "one thousand, five"),
    (1234, "one thousand, two hundred thirty-four"), (12345, "twelve thousand, three hundred forty-five"),
    (1000000, "one million"), (2500000, "two million, five hundred thousand"),
])
def test_cardinals_without_and(n, words):
    assert T.number_to_words(n, andword="") == words


def test_cardinals_with_and_years_and_ordinals():
    assert T.number_to_words(101) == "one hundred and one"
    assert T.number_to_words(1005) == "one thousand and five"
    assert T.number_to_words(1234) == "one thousand, two hundred and thirty-four"
    assert T.number_to_words(1984, andword="", zero="oh", group=2) == "nineteen, eighty-four"
    assert T.number_to_words(1905, andword="", zero="oh", group=2) == "nineteen, oh five"
    for s, w in [("1st", "first"), ("2nd", "second"), ("3rd", "third"), ("4th", "fourth"), ("5th", "fifth"), ("8th", "eighth"),
                 ("9th", "ninth"), ("12th", "twelfth"), ("20th", "twentieth"), ("21st", "twenty-first"), ("100th", "one hundredth"),
                 ("101st", "one hundred and first"), ("1000th", "one thousandth")]:
        assert T.ordinal_words(s) == w


@pytest.mark.parametrize("text,out", [
    ("1984", "nineteen eighty-four"), ("2000", "two thousand"), ("2005", "two thousand five"), ("1900", "nineteen hundred"),
    ("2010", "twenty ten"), ("3000", "three thousand"), ("1,000", "one thousand"), ("3.14", "three point fourteen"),
    ("$250", "two hundred fifty dollars"), ("$1", "one dollar"), ("$0.05", "five cents"), ("$2.50", "two dollars, fifty cents"),
    ("£20", "twenty pounds"), ("the 3rd", "the third"),
])
def test_normalize_numbers(text, out):                     # texts.py:126-133
    assert T.normalize_numbers(text) == out


def test_english_cleaners():
    assert T.english_cleaners("Mr. Smith  and Dr. Jones\tSt. Louis") == "mister smith and doctor jones saint louis"
    assert T.english_cleaners("Café — “déjà vu”") == 'cafe -- "deja vu"'
    assert T.english_cleaners("In 1984, he paid $1,250.50 for 2 tables.") == \
        "in nineteen eighty-four, he paid twelve fifty dollars, fifty cents for two tables."
    assert T.basic_cleaners("A  B\nC") == "a b c" and T.transliteration_cleaners("Ünï  X") == "uni x"


def test_text_to_array_and_padding():
    assert len(LJHPS.Texts.characters) == LJHPS.Encoder.Transformer.vocab_size == 43          # hparams.py:264,293
    assert len(DataBakerHPS.Texts.characters) == DataBakerHPS.Encoder.Transformer.vocab_size == 39
    a = T.text_to_array("Hello, world!", LJHPS)
    chars = LJHPS.Texts.characters
    assert "".join(chars[i] for i in a) == "^hello, world!~" and a[0] == 1 and a[-1] == 2
    with pytest.raises(KeyError):
        T.text_to_array("a # b", LJHPS)                    # '#' is not in the symbol table: KeyError, as in the reference
    b = T.pinyin_to_array("Ni3 hao3", DataBakerHPS)
    assert "".join(DataBakerHPS.Texts.characters[i] for i in b) == "^ni3 hao3~"
    ids, lens = T.pad_batch([a, b])
    assert ids.shape == (2, len(a)) and ids.dtype == np.int32 and list(lens) == [len(a), len(b)] and (ids[1, len(b):] == 0).all()


def test_testutils_figures(tmp_path):
    """The harnesses' figures (reference audio/utils.py:42-116, called from train.py:319-323 / inference.py:160-164): one PDF per utterance
    for the predicted mels, one per utterance and decoder block for the alignments (3-D: one panel, 4-D: one panel per head)."""
    import os
    import numpy as np
    import pytest
    pytest.importorskip("matplotlib")
    from vaenar_tts_amd.audio.utils import TestUtils
    from vaenar_tts_amd.configs import LJHPS
    t = TestUtils.__new__(TestUtils)                      # (no engine: the figures are host-only)
    t.hps, t.save_dir = LJHPS, str(tmp_path)
    r = np.random.default_rng(0)
    mels = r.standard_normal((2, 30, 80)).astype(np.float32)
    t.draw_melspectrograms(7, mels, [30, 21], [b"utt-a", "utt-b"], prefix="test")
    ali4 = r.random((2, 4, 15, 9)).astype(np.float32)
    texts = r.integers(3, 29, (2, 9))
    t.multi_draw_attention_alignments(ali4, texts, [9, 6], [15, 11], 7, ["utt-a", "utt-b"], "test-decoder-attention-0")
    t.multi_draw_attention_alignments(ali4[:, 0], texts, [9, 6], [15, 11], 7, ["utt-a", "utt-b"], "post")
    names = sorted(os.listdir(tmp_path))
    assert names == ["post-utt-a-7.pdf", "post-utt-b-7.pdf", "test-decoder-attention-0-utt-a-7.pdf", "test-decoder-attention-0-utt-b-7.pdf",
                     "test-utt-a-7.pdf", "test-utt-b-7.pdf"], names
    assert all(os.path.getsize(os.path.join(tmp_path, n)) > 1000 for n in names)
    assert t._ids_to_symbols([3, 4, 5]) == list(LJHPS.Texts.characters[3:6])
