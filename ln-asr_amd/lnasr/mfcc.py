"""MFCC feature extraction with the reference's class surface (lnasr/mfcc.py:16-175).

`calc_mfcc` runs the whole chain on the GPU (csrc/mfcc.hip): pre-emphasis, framing with a
zero-padded tail, Hamming window, rFFT, power, mel filter bank, 20 log10, utterance mean
normalisation, DCT-II (ortho), a second mean normalisation of cepstra 1..12 (visible in the
returned cepstrum too, as in the reference), log-energy, delta and delta-delta: D = 39.

The Hamming window and the mel filter bank are model constants built once on the host
(the reference rebuilds the bank on every call, mfcc.py:145).

Extras: `calc_mfcc_batch` (many utterances, float64 or int16 PCM, one launch sequence).
"""
import numpy as np

from . import _ffi
from .utils import create_hamming


class MFCC:
    D = 39

    def __init__(self, fs=16000, frame_T=25e-3, frame_stride=10e-3, hpf_alpha=0.97, fft_N=512,
                 mfbank_num=40, cepstrum_num=12, *, ctx=None):
        self.fs = fs
        self.frame_T = frame_T
        self.frame_L = int(self.fs * self.frame_T)               # samples per frame
        self.frame_stride = frame_stride
        self.frame_step = int(self.fs * self.frame_stride)       # hop in samples
        self.hpf_alpha = hpf_alpha
        self.fft_N = fft_N
        self.fft_size = self.fft_N // 2 + 1
        self.mfbank_num = mfbank_num
        self.cepstrum_num = cepstrum_num
        self._ctx = ctx
        self._consts = None

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _ffi.default_ctx()
        return self._ctx

    def calc_high_pass_filter(self, signal: np.ndarray) -> np.ndarray:
        """y(t) = x(t) - alpha x(t-1), y(0) = x(0)  (mfcc.py:50-64)."""
        return self.ctx.preemphasis(np.asarray(signal, dtype=np.float64), self.hpf_alpha)

    def calc_mel(self, hz: np.ndarray):
        """Hertz -> mel with a NATURAL log, as the reference has it (mfcc.py:66-72)."""
        return 2595 * np.log(1 + hz / 700)

    def calc_mel_inv(self, mel: np.ndarray):
        return 700 * (np.exp(mel / 2595) - 1)

    def create_filter_bank(self) -> np.ndarray:
        """[mfbank_num, fft_N/2+1] unnormalised triangles on mel-equispaced FFT bins (mfcc.py:78-106)."""
        M, N = self.mfbank_num, self.fft_N
        edges_hz = self.calc_mel_inv(np.linspace(self.calc_mel(0), self.calc_mel(self.fs / 2), M + 2))
        edge = np.floor((N / self.fs) * edges_hz)
        bank = np.zeros((M, N // 2 + 1))
        for k in range(M):
            lo, mi, hi = int(edge[k]), int(edge[k + 1]), int(edge[k + 2])
            bank[k, lo:mi] = (np.arange(lo, mi) - lo) / (mi - lo)
            bank[k, mi:hi] = (hi - np.arange(mi, hi)) / (hi - mi)
        return bank

    def _cfg(self):
        key = (self.fs, self.frame_L, self.frame_step, self.fft_N, self.mfbank_num, self.cepstrum_num, self.hpf_alpha)
        if self._consts is None or self._consts[0] != key:
            cfg = _ffi.MfccCfgStruct(int(self.fs), self.frame_L, self.frame_step, int(self.fft_N),
                                     int(self.mfbank_num), int(self.cepstrum_num), float(self.hpf_alpha))
            self._consts = (key, cfg, create_hamming(self.frame_L), self.create_filter_bank())
        return self._consts[1:]

    def calc_mfcc(self, signal: np.ndarray):
        """signal [n] -> (frames_power [Nf,257], frames_cepstrum [Nf,40], features [Nf,39])  (mfcc.py:108-175)."""
        cfg, win, bank = self._cfg()
        signal = np.asarray(signal)
        if signal.dtype != np.int16:
            signal = np.asarray(signal, dtype=np.float64)
        power, cep, feat, _ = self.ctx.mfcc(signal, np.array([0, len(signal)], dtype=np.int64), cfg, win, bank)
        return (power, cep, feat)

    def calc_mfcc_batch(self, signals, offsets=None, want_power=False, want_cepstrum=False):
        """Either a list of 1-D arrays, or one concatenated array (host or DeviceArray) + offsets [U+1].
        Returns (features [sumNf, 39], frame_offsets [U+1]) (+ power / cepstrum when asked)."""
        cfg, win, bank = self._cfg()
        if offsets is None:
            lens = [len(s) for s in signals]
            offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            signals = np.concatenate([np.asarray(s) for s in signals])
        power, cep, feat, foff = self.ctx.mfcc(signals, offsets, cfg, win, bank, want_power, want_cepstrum)
        out = (feat, foff)
        if want_power:
            out += (power,)
        if want_cepstrum:
            out += (cep,)
        return out
