"""Vocoder step after the text->mel path: counterpart of the reference package `audio/` (audio/__init__.py:1-2)."""
from .audio import Audio
from .utils import TestUtils

__all__ = ["Audio", "TestUtils"]
