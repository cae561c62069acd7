"""Solution strings and best-so-far tracking -- the host side of
rlsolver/methods/util_evaluator.py:22-150 (EncoderBase64, Evaluator).

EncoderBase64 keeps the reference's text format bit for bit (digits ``0-9A-Za-z_$``, most
significant spin first, zero-filled to ceil(N/6) characters, 120-column wrapping) but converts
through ``int.from_bytes`` / ``np.packbits`` instead of building decimal strings, so a 10^6-spin
solution encodes in milliseconds.
"""
from __future__ import annotations

import os
import time
from typing import Union

import numpy as np
import torch as th

TEN = th.Tensor
_DIGITS = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz_$"
_VALUE = {c: i for i, c in enumerate(_DIGITS)}


class EncoderBase64:
    def __init__(self, encode_len: int):
        self.encode_len = encode_len
        self.string_len = -(-encode_len // 6)
        self.base_digits = _DIGITS
        self.base_num = 64

    def bool_to_str(self, x_bool) -> str:
        bits = np.asarray(x_bool.detach().cpu().numpy() if isinstance(x_bool, th.Tensor) else x_bool).astype(np.uint8)
        n = bits.shape[0]
        pad = (-n) % 8
        packed = np.packbits(np.concatenate([np.zeros(pad, np.uint8), bits]))   # MSB-first, value-preserving
        x_int = int.from_bytes(packed.tobytes(), "big")
        chars = []
        while True:
            chars.append(_DIGITS[x_int & 63])
            x_int >>= 6
            if x_int == 0:
                break
        x_str = "".join(reversed(chars))
        if len(x_str) > 120:
            x_str = "\n".join(x_str[i:i + 120] for i in range(0, len(x_str), 120))
        if len(x_str) > 64:
            x_str = "\n" + x_str
        return x_str.zfill(self.string_len)

    def str_to_bool(self, x_str: str) -> TEN:
        x_int = 0
        for ch in x_str:
            if ch in "\n ":
                continue
            x_int = (x_int << 6) | _VALUE[ch]
        nbytes = (self.encode_len + 7) // 8
        if x_int >> (8 * nbytes):
            raise ValueError("string encodes more bits than encode_len")
        bits = np.unpackbits(np.frombuffer(x_int.to_bytes(nbytes, "big"), dtype=np.uint8))
        return th.from_numpy(bits[-self.encode_len:].astype(bool))


class Evaluator:
    """Best-solution tracker with the reference's interface (util_evaluator.py:68-150).  record2
    picks the batch best on the device; like the reference it returns a Python bool (one scalar
    host read per call -- call it at episode boundaries, not per step)."""

    def __init__(self, save_dir: str, num_bits: int, x: TEN, v: float, if_maximize: bool):
        self.start_timer = time.time()
        self.recorder1 = []
        self.recorder2 = []
        enc = EncoderBase64(encode_len=num_bits)
        self.bool_to_str = enc.bool_to_str
        self.str_to_bool = enc.str_to_bool
        self.best_x = x
        self.best_v = v
        self.if_maximize = if_maximize
        self.save_dir = save_dir
        os.makedirs(self.save_dir, exist_ok=True)
        self.record1(i=0, v=self.best_v)
        self.record2(i=0, vs=self.best_v, xs=self.best_x)

    def record1(self, i: float, v: float):
        self.recorder1.append((i, v))

    def record2(self, i: float, vs: Union[TEN, float], xs: TEN):
        if len(xs.shape) == 2:
            good_i = vs.argmax() if self.if_maximize else vs.argmin()
            good_x, good_v = xs[good_i], vs[good_i]
        else:
            good_x, good_v = xs, vs
        good_v = float(good_v)
        self.recorder2.append((i, good_v, time.time() - self.start_timer))
        if_update = (good_v > self.best_v) if self.if_maximize else (good_v < self.best_v)
        if if_update:
            self.best_x = good_x
            self.best_v = good_v
        return if_update

    def save_record_draw_plot(self, fig_dpi: int = 300):
        if not self.recorder1 or not self.recorder2:
            return
        np.save(f"{self.save_dir}/recorder1.npy", np.array(self.recorder1))
        np.save(f"{self.save_dir}/recorder2.npy", np.array(self.recorder2))

    def logging_print(self, show_str: str = '', if_show_x: bool = False):
        used_time = int(time.time() - self.start_timer)
        x_str = self.best_x_str if if_show_x else ''
        i = self.recorder2[-1][0]
        log_str = f"|{i:6} {used_time:4} sec  best {self.best_v:12.4f} {show_str}  x_str: {x_str}"
        print(log_str, flush=True)
        return log_str

    @property
    def first_v(self) -> float:
        return self.recorder2[0][1]

    @property
    def best_x_str(self):
        return self.bool_to_str(self.best_x).replace('\n', '')
