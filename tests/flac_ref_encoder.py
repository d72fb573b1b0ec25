"""Bit-level FLAC ENCODER for the tests of the native decoder (csrc/ou_flac.cpp): test infrastructure, written from the published
format like the decoder, but as its mirror image and with every feature selectable -- subframe types, predictor orders, Rice
partition orders / parameter widths / escapes, wasted bits, stereo decorrelation, block-size and sample-rate codes, blocking
strategy, extra metadata, an ID3v2 prefix.  Slow (pure Python): small signals only."""
import hashlib


class BitWriter:
    def __init__(self):
        self.acc, self.n = 0, 0

    def put(self, value, bits):
        if bits:
            self.acc = (self.acc << bits) | (value & ((1 << bits) - 1))
            self.n += bits

    def sput(self, value, bits):
        assert -(1 << (bits - 1)) <= value < (1 << (bits - 1)), (value, bits)
        self.put(value, bits)

    def unary(self, q):
        self.put(1, q + 1)  # q zeros, then a one

    def align(self):
        if self.n % 8:
            self.put(0, 8 - self.n % 8)

    def bytes(self):
        assert self.n % 8 == 0
        return self.acc.to_bytes(self.n // 8, "big") if self.n else b""


def crc8(b):
    c = 0
    for x in b:
        c ^= x
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(b):
    c = 0
    for x in b:
        c ^= x << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def utf8_number(n):
    if n < 0x80:
        return bytes([n])
    nb = 2 if n < 0x800 else 3 if n < 0x10000 else 4 if n < 0x200000 else 5 if n < 0x4000000 else 6 if n < 0x80000000 else 7
    tail = []
    for _ in range(nb - 1):
        tail.append(0x80 | (n & 0x3F))
        n >>= 6
    return bytes([((0xFF << (8 - nb)) & 0xFF) | n] + tail[::-1])


def _residual(w, res, bs, order, po, method, escape_parts=(), ks=None):
    pbits, esc = (4, 15) if method == 0 else (5, 31)
    w.put(method, 2)
    w.put(po, 4)
    parts = 1 << po
    i = 0
    for pt in range(parts):
        cnt = (bs >> po) - (order if pt == 0 else 0) if po else bs - order
        seg = res[i:i + cnt]
        i += cnt
        if pt in escape_parts:
            nb = max([1] + [v.bit_length() + 1 for v in seg])
            w.put(esc, pbits)
            w.put(nb, 5)
            for v in seg:
                w.sput(v, nb)
            continue
        us = [2 * v if v >= 0 else -2 * v - 1 for v in seg]
        if ks is not None:
            k = ks[pt % len(ks)]
        else:
            mean = sum(us) / max(1, len(us))
            k = min(esc - 1, max(0, int(mean + 1).bit_length() - 1))
        w.put(k, pbits)
        for u in us:
            w.unary(u >> k)
            w.put(u & ((1 << k) - 1), k)
    assert i == len(res)


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def subframe(w, x, bps, spec):
    """spec: ("constant",) | ("verbatim",) | ("fixed", order, po, method, escape_parts, ks) | ("lpc", coefs, precision, shift, po,
    method, escape_parts, ks); optional last element {"wasted": k}"""
    opts = spec[-1] if isinstance(spec[-1], dict) else {}
    if opts:
        spec = spec[:-1]
    wasted = opts.get("wasted", 0)
    if wasted:
        assert all(v % (1 << wasted) == 0 for v in x)
        x = [v >> wasted for v in x]
        bps -= wasted
    bs = len(x)
    kind = spec[0]
    w.put(0, 1)
    if kind == "constant":
        assert len(set(x)) == 1
        w.put(0, 6)
    elif kind == "verbatim":
        w.put(1, 6)
    elif kind == "fixed":
        w.put(8 + spec[1], 6)
    else:
        w.put(31 + len(spec[1]), 6)
    if wasted:
        w.put(1, 1)
        w.unary(wasted - 1)
    else:
        w.put(0, 1)
    if kind == "constant":
        w.sput(x[0], bps)
    elif kind == "verbatim":
        for v in x:
            w.sput(v, bps)
    elif kind == "fixed":
        order, po, method = spec[1], spec[2], spec[3]
        esc = spec[4] if len(spec) > 4 else ()
        ks = spec[5] if len(spec) > 5 else None
        c = FIXED[order]
        for v in x[:order]:
            w.sput(v, bps)
        res = [x[i] - sum(c[j] * x[i - 1 - j] for j in range(order)) for i in range(order, bs)]
        _residual(w, res, bs, order, po, method, esc, ks)
    else:
        coefs, prec, shift, po, method = spec[1], spec[2], spec[3], spec[4], spec[5]
        esc = spec[6] if len(spec) > 6 else ()
        ks = spec[7] if len(spec) > 7 else None
        order = len(coefs)
        for v in x[:order]:
            w.sput(v, bps)
        w.put(prec - 1, 4)
        w.sput(shift, 5)
        for cf in coefs:
            w.sput(cf, prec)
        res = [x[i] - (sum(coefs[j] * x[i - 1 - j] for j in range(order)) >> shift) for i in range(order, bs)]
        _residual(w, res, bs, order, po, method, esc, ks)


BLOCK_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
RATE_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
SIZE_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def frame(chans, bps, fs, number, specs, stereo=None, variable=False, rate_mode="streaminfo", size_from_streaminfo=False,
          force_explicit_block=False):
    """chans: list of per-channel int lists (already L / R); stereo: None | 8 | 9 | 10"""
    bs = len(chans[0])
    nch = len(chans)
    hdr = bytearray([0xFF, 0xF8 | (1 if variable else 0)])
    bcode = BLOCK_CODES.get(bs) if not force_explicit_block else None
    extra = b""
    if bcode is None:
        if bs <= 256:
            bcode, extra = 6, bytes([bs - 1])
        else:
            bcode, extra = 7, (bs - 1).to_bytes(2, "big")
    rextra = b""
    if rate_mode == "streaminfo":
        rcode = 0
    elif rate_mode == "table":
        rcode = RATE_CODES[fs]
    elif rate_mode == "khz8":
        rcode, rextra = 12, bytes([fs // 1000])
    elif rate_mode == "hz16":
        rcode, rextra = 13, fs.to_bytes(2, "big")
    else:
        rcode, rextra = 14, (fs // 10).to_bytes(2, "big")
    hdr.append((bcode << 4) | rcode)
    chc = stereo if stereo is not None else nch - 1
    hdr.append((chc << 4) | ((0 if size_from_streaminfo else SIZE_CODES[bps]) << 1))
    hdr += utf8_number(number)
    hdr += extra + rextra
    hdr.append(crc8(hdr))
    w = BitWriter()
    if stereo is None:
        coded = [(c, bps) for c in chans]
    else:
        L, R = chans
        side = [a - b for a, b in zip(L, R)]
        if stereo == 8:
            coded = [(L, bps), (side, bps + 1)]
        elif stereo == 9:
            coded = [(side, bps + 1), (R, bps)]
        else:
            coded = [([(a + b) >> 1 for a, b in zip(L, R)], bps), (side, bps + 1)]
    for (c, b), sp in zip(coded, specs):
        subframe(w, list(c), b, sp)
    w.align()
    fr = bytes(hdr) + w.bytes()
    return fr + crc16(fr).to_bytes(2, "big")


def stream(chans, bps, fs, frames, total=None, md5=True, id3=False, extra_blocks=(), min_block=16, max_block=65535):
    """chans: (channels x T) ints -> bytes.  frames: list of frame byte strings."""
    nch, n = len(chans), len(chans[0])
    nb = (bps + 7) // 8
    raw = bytearray()
    for i in range(n):
        for c in range(nch):
            raw += (chans[c][i] & ((1 << (8 * nb)) - 1)).to_bytes(nb, "little")
    digest = hashlib.md5(bytes(raw)).digest() if md5 else bytes(16)
    si = bytearray()
    si += min_block.to_bytes(2, "big") + max_block.to_bytes(2, "big") + (0).to_bytes(3, "big") + (0).to_bytes(3, "big")
    v = (fs << 44) | ((nch - 1) << 41) | ((bps - 1) << 36) | (n if total is None else total)
    si += v.to_bytes(8, "big") + digest
    out = bytearray()
    if id3:
        body = b"\x00" * 37
        out += b"ID3\x04\x00\x00" + bytes([0, 0, 0, len(body)]) + body
    out += b"fLaC"
    blocks = [(0, bytes(si))] + list(extra_blocks)
    for i, (t, body) in enumerate(blocks):
        out += bytes([(0x80 if i == len(blocks) - 1 else 0) | t]) + len(body).to_bytes(3, "big") + body
    return bytes(out) + b"".join(frames)
