"""A small lossless-JPEG (ITU-T T.81 SOF3) ENCODER for the ingest tests: the counterpart of librawdev's decoder, written
independently from the standard's text (Annex H for the predictors and the difference categories, Annex C for the canonical
Huffman codes, B.2 for the segments).  Pure Python, slow, small images only."""
import struct

import numpy as np

# code lengths for the 17 difference categories SSSS = 0..16 (Kraft sum < 1; no code is all ones)
_LENGTHS = {0: 2, 1: 3, 2: 3, 3: 3, 4: 3, 5: 3, 6: 4, 7: 5, 8: 6, 9: 7, 10: 8, 11: 9, 12: 10, 13: 11, 14: 12, 15: 13, 16: 14}


def _canonical(lengths):
    """{symbol: length} -> (counts[16], symbols in code order, {symbol: (code, length)}) as DHT stores them (C.2)."""
    order = sorted(lengths, key=lambda s: (lengths[s], s))
    counts = [0] * 16
    for s in order:
        counts[lengths[s] - 1] += 1
    codes, code, prev = {}, 0, lengths[order[0]]
    for s in order:
        code <<= lengths[s] - prev
        prev = lengths[s]
        codes[s] = (code, lengths[s])
        code += 1
    return counts, order, codes


class _Bits:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, value, nbits):
        self.acc = (self.acc << nbits) | (value & ((1 << nbits) - 1))
        self.n += nbits
        while self.n >= 8:
            byte = (self.acc >> (self.n - 8)) & 0xff
            self.out.append(byte)
            if byte == 0xff:
                self.out.append(0x00)                    # byte stuffing (B.1.1.5)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)     # pad with ones (F.1.2.3)


def encode(samples, components=1, precision=16, predictor=1, point_transform=0, restart_rows=0):
    """samples: (height, width * components) integer array, components interleaved along a row.  restart_rows > 0 puts a
    restart marker every that many rows (an interval must be whole lines, H.1.2.1)."""
    a = np.asarray(samples).astype(np.int64) >> point_transform
    h, wn = a.shape
    assert wn % components == 0
    w = wn // components
    counts, order, codes = _canonical(_LENGTHS)
    out = bytearray(b"\xff\xd8")
    out += b"\xff\xc4" + struct.pack(">H", 2 + 1 + 16 + len(order)) + bytes([0x00]) + bytes(counts) + bytes(order)
    out += b"\xff\xc3" + struct.pack(">HBHHB", 8 + 3 * components, precision, h, w, components)
    for c in range(components):
        out += bytes([c + 1, 0x11, 0])
    if restart_rows:
        out += b"\xff\xdd" + struct.pack(">HH", 4, restart_rows * w)
    out += b"\xff\xda" + struct.pack(">HB", 6 + 2 * components, components)
    for c in range(components):
        out += bytes([c + 1, 0x00])
    out += bytes([predictor, 0, point_transform])
    bits = _Bits()
    init = 1 << (precision - point_transform - 1)
    rst = 0
    for y in range(h):
        if restart_rows and y and y % restart_rows == 0:
            bits.flush()
            out += bits.out + bytes([0xff, 0xd0 + rst % 8])
            rst += 1
            bits = _Bits()
        first_line = y == 0 or (restart_rows and y % restart_rows == 0)
        for x in range(w):
            for c in range(components):
                v = int(a[y, x * components + c])
                if first_line and x == 0:
                    px = init
                elif first_line:
                    px = int(a[y, (x - 1) * components + c])
                elif x == 0:
                    px = int(a[y - 1, c])
                else:
                    ra, rb, rc = (int(a[y, (x - 1) * components + c]), int(a[y - 1, x * components + c]),
                                  int(a[y - 1, (x - 1) * components + c]))
                    px = [None, ra, rb, rc, ra + rb - rc, ra + ((rb - rc) >> 1), rb + ((ra - rc) >> 1), (ra + rb) >> 1][predictor]
                diff = (v - px) & 0xffff                       # modulo 2^16 (H.1.2.1)
                if diff >= 32768:
                    diff -= 65536                              # ... represented in -32767 .. 32768
                if diff == -32768:
                    diff = 32768
                ssss = 0 if diff == 0 else int(abs(diff)).bit_length()
                code, length = codes[ssss]
                bits.put(code, length)
                if 0 < ssss < 16:
                    bits.put(diff if diff > 0 else diff + (1 << ssss) - 1, ssss)
    bits.flush()
    out += bits.out + b"\xff\xd9"
    return bytes(out)
