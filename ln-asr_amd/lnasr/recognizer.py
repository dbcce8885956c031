"""Recognizer shell with the reference's constructor wiring (lnasr/recognizer.py:20-48).

The reference's `recognize()` is unimplemented (`pass`) and its language-model / lexicon side
is out of scope here.  `AcousticModel` wires MFCC + GMMHMM exactly as the
reference does; `Recognizer.recognize` runs the acoustic path the reference stops short of:
PCM -> MFCC -> Viterbi state path (parity unpinned: nothing to compare with upstream).
"""
import numpy as np

from .mfcc import MFCC
from .gmmhmm import GMMHMM


class AcousticModel:
    def __init__(self, modelfile):
        self.mfcc = MFCC()
        self.gmmhmm = GMMHMM(n=2, m=2, d=self.mfcc.D)
        self.gmmhmm.load(modelfile)


class LanguageModel:
    """The n-gram language model is not part of the accelerated path; the model file is only recorded."""

    def __init__(self, modelfile):
        self.modelfile = modelfile
        self.ngram = None


class Recognizer:
    def __init__(self, model: dict):
        self.am = AcousticModel(model["am"])
        self.lm = LanguageModel(model["lm"]) if model.get("lm") is not None else None

    def recognize(self, data: np.ndarray):
        """PCM samples -> most likely HMM state sequence (uint32 [Nf])."""
        _, _, feats = self.am.mfcc.calc_mfcc(np.asarray(data))
        return self.am.gmmhmm.decode(feats)
