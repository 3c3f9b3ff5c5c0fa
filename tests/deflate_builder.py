"""Hand-built zlib streams (RFC 1950 / 1951) for the inflate tests: a tiny bit writer that emits stored, fixed and dynamic
blocks from explicit code-length sets - legal ones the stock encoder never produces (a one-symbol distance code, a block
that holds only the end-of-block symbol, repeat codes running across the literal / distance boundary) and illegal ones
(over-subscribed or incomplete sets, missing end-of-block code, invalid symbols, distances before the start).  The verdict
of the reference's own zlib (`uncompress`, oracle/_ref) is the yardstick; tests compare the CPU build of
c-blosc_amd/csrc/inflate_serial.h and the GPU kernel with it."""
import zlib

import numpy as np

LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DIST_EXTRA = [0, 0, 0, 0] + [e for e in range(1, 14) for _ in (0, 1)]
CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


class BitWriter:
    def __init__(self):
        self.acc = 0; self.n = 0; self.out = bytearray()

    def bits(self, v, k):          # LSB first
        self.acc |= (v & ((1 << k) - 1)) << self.n; self.n += k
        while self.n >= 8:
            self.out.append(self.acc & 0xff); self.acc >>= 8; self.n -= 8

    def code(self, c, k):          # Huffman codes go in MSB first
        r = 0
        for i in range(k):
            r = (r << 1) | ((c >> i) & 1)
        self.bits(r, k)

    def align(self):
        if self.n:
            self.bits(0, 8 - self.n)

    def raw(self, b):
        self.align(); self.out += bytes(b)


def canonical(lens):
    """symbol -> (code, length) for the symbols with a length; works for incomplete sets too (codes are what a decoder that
    assigns canonically would expect)"""
    mx = max(lens) if lens else 0
    cnt = [0] * (mx + 2)
    for l in lens:
        if l:
            cnt[l] += 1
    nxt = [0] * (mx + 2); code = 0
    for l in range(1, mx + 1):
        code = (code + cnt[l - 1]) << 1 if l > 1 else 0
        nxt[l] = code
    out = {}
    for s, l in enumerate(lens):
        if l:
            out[s] = (nxt[l], l); nxt[l] += 1
    return out


def len_symbol(length):
    for c in range(28, -1, -1):
        if length >= LEN_BASE[c] and (c == 28 or length < LEN_BASE[c] + (1 << LEN_EXTRA[c])):
            if c == 28 and length != 258:
                continue
            return 257 + c, length - LEN_BASE[c], LEN_EXTRA[c]
    raise ValueError(length)


def dist_symbol(dist):
    for c in range(29, -1, -1):
        if dist >= DIST_BASE[c]:
            return c, dist - DIST_BASE[c], DIST_EXTRA[c]
    raise ValueError(dist)


def emit_ops(w, ops, lit, dst):
    """ops: ints (literal bytes), ('m', length, distance), ('sym', literal/length symbol) or ('dsym', length, distance symbol,
    extra bits) for symbols no encoder would write"""
    for o in ops:
        if isinstance(o, int):
            w.code(*lit[o])
        elif o[0] == 'm':
            s, ev, eb = len_symbol(o[1]); w.code(*lit[s]); w.bits(ev, eb)
            d, dv, db = dist_symbol(o[2]); w.code(*dst[d]); w.bits(dv, db)
        elif o[0] == 'sym':
            w.code(*lit[o[1]])
        elif o[0] == 'dsym':
            s, ev, eb = len_symbol(o[1]); w.code(*lit[s]); w.bits(ev, eb)
            w.code(*dst[o[2]]); w.bits(o[3], DIST_EXTRA[o[2]] if o[2] < 30 else 0)
    w.code(*lit[256])


def cl_sequence(lens):
    """code-length alphabet symbols for a list of lengths, plain (no repeat codes)"""
    return [(l, 0, 0) for l in lens]


def dynamic_block(w, final, litlens, distlens, ops, clseq=None, cllens=None, hlit=None, hdist=None):
    """one dynamic block.  clseq: explicit [(symbol, extra value, extra bits)] for the code-length alphabet (default: one
    symbol per length); cllens: the 19 code lengths of that alphabet (default: 5 bits for every used symbol, padded so
    the set is complete)"""
    seq = clseq if clseq is not None else cl_sequence(list(litlens) + list(distlens))
    if cllens is None:
        used = sorted({s for s, _, _ in seq})
        cllens = [0] * 19
        # a complete set: give the used symbols lengths from a canonical shape of size len(used)
        k = len(used)
        if k == 1:
            cllens[used[0]] = 1; cllens[(used[0] + 1) % 19] = 1
        else:
            import math
            lo = int(math.floor(math.log2(k))); n_long = 2 * (k - (1 << lo)); n_short = k - n_long
            for i, s in enumerate(used):
                cllens[s] = lo if i < n_short else lo + 1
    cl = canonical(cllens)
    w.bits(1 if final else 0, 1); w.bits(2, 2)
    w.bits((hlit if hlit is not None else len(litlens)) - 257, 5); w.bits((hdist if hdist is not None else len(distlens)) - 1, 5)
    ncode = 19
    while ncode > 4 and cllens[CL_ORDER[ncode - 1]] == 0:
        ncode -= 1
    w.bits(ncode - 4, 4)
    for i in range(ncode):
        w.bits(cllens[CL_ORDER[i]], 3)
    for s, ev, eb in seq:
        w.code(*cl[s]); w.bits(ev, eb)
    emit_ops(w, ops, canonical(list(litlens)), canonical(list(distlens)))


def fixed_block(w, final, ops):
    lit = canonical([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8); dst = canonical([5] * 32)
    w.bits(1 if final else 0, 1); w.bits(1, 2)
    emit_ops(w, ops, lit, dst)


def stored_block(w, final, data, nlen_xor=0xffff):
    w.bits(1 if final else 0, 1); w.bits(0, 2); w.align()
    n = len(data)
    w.out += bytes([n & 0xff, n >> 8, (n ^ nlen_xor) & 0xff, ((n ^ nlen_xor) >> 8) & 0xff]) + bytes(data)


def plain_of(ops_list):
    out = bytearray()
    for ops in ops_list:
        for o in ops:
            if isinstance(o, int):
                out.append(o)
            elif o[0] in ('m', 'dsym') and o[0] == 'm':
                for _ in range(o[1]):
                    out.append(out[-o[2]])
    return bytes(out)


def finish(w, plain, cmf=0x78, flg=0x9c, adler=None):
    w.align()
    a = zlib.adler32(plain) if adler is None else adler
    return np.frombuffer(bytes([cmf, flg]) + bytes(w.out) + a.to_bytes(4, 'big'), np.uint8).copy()


def cases():
    """(name, stream, plain length the caller should offer) - verdicts come from the reference at test time"""
    out = []
    lit_ab = [0] * 257; lit_ab[65] = 2; lit_ab[66] = 2; lit_ab[67] = 2; lit_ab[256] = 3
    lit_ab_len = list(lit_ab) + [3]                      # + symbol 257 (length 3): complete (2,2,2,3,3)
    ops = [65, 66, 67, 65, ('m', 3, 3), ('m', 3, 1)]
    # 1. one-symbol distance code (incomplete, longest code 1 bit): legal
    w = BitWriter(); dynamic_block(w, True, lit_ab_len, [0, 0, 1], [65, 66, 67, ('m', 3, 3), ('m', 3, 3)])
    out.append(("dist-one-symbol", finish(w, plain_of([[65, 66, 67, ('m', 3, 3), ('m', 3, 3)]])), 9))
    # 2. no distance code at all, literals only
    lit_only = [0] * 257; lit_only[65] = 1; lit_only[256] = 1
    w = BitWriter(); dynamic_block(w, True, lit_only, [0], [65, 65, 65])
    out.append(("no-dist-code", finish(w, b"AAA"), 3))
    # 3. block with only end-of-block (single literal/length code of one bit), then a stored block
    eob_only = [0] * 257; eob_only[256] = 1
    w = BitWriter(); dynamic_block(w, False, eob_only, [0], []); stored_block(w, True, b"xyz")
    out.append(("eob-only-then-stored", finish(w, b"xyz"), 3))
    # 4. incomplete literal/length code with a 2-bit longest code: illegal
    bad = [0] * 257; bad[65] = 2; bad[256] = 2
    w = BitWriter(); dynamic_block(w, True, bad, [0], [65])
    out.append(("lit-incomplete", finish(w, b"A"), 1))
    # 5. over-subscribed literal/length code
    bad = [0] * 257; bad[65] = 1; bad[66] = 1; bad[256] = 1
    w = BitWriter(); dynamic_block(w, True, bad, [0], [65])
    out.append(("lit-oversubscribed", finish(w, b"A"), 1))
    # 6. missing end-of-block code
    bad = [0] * 257; bad[65] = 1; bad[66] = 1
    w = BitWriter(); bad2 = list(bad); bad2[256] = 0
    w.bits(1, 1); w.bits(2, 2); w.bits(0, 5); w.bits(0, 5)
    cll = [0] * 19; cll[0] = 1; cll[1] = 1; clc = canonical(cll)
    nc = 19
    while nc > 4 and cll[CL_ORDER[nc - 1]] == 0:
        nc -= 1
    w.bits(nc - 4, 4)
    for i in range(nc):
        w.bits(cll[CL_ORDER[i]], 3)
    for l in bad2 + [0]:
        w.code(*clc[l])
    out.append(("no-eob-code", finish(w, b""), 4))
    # 7. incomplete code-length code
    w = BitWriter(); cll = [0] * 19; cll[0] = 2; cll[1] = 2; cll[2] = 2
    dynamic_block(w, True, lit_only, [0], [65], cllens=cll)
    out.append(("cl-incomplete", finish(w, b"A"), 1))
    # 8. repeat code 16 with nothing before it
    w = BitWriter(); seq = [(16, 0, 2)] + cl_sequence(lit_only[3:] + [0])
    dynamic_block(w, True, lit_only, [0], [65], clseq=seq)
    out.append(("repeat-at-start", finish(w, b"A"), 1))
    # 9. zero run (18) that crosses from the literal/length lengths into the distance lengths: legal
    ll = [0] * 257; ll[65] = 1; ll[256] = 1
    seq = cl_sequence(ll[:66]) + [(18, 138 - 11, 7), (18, (256 - 66 - 138) - 11, 7), (1, 0, 0), (18, 30 - 11, 7)]
    w = BitWriter(); dynamic_block(w, True, ll, [0] * 30, [65, 65], clseq=seq, hlit=257, hdist=30)
    out.append(("zero-run-across-boundary", finish(w, b"AA"), 2))
    # 10. repeat running past the last length
    seq = cl_sequence(ll[:66]) + [(18, 138 - 11, 7), (18, (256 - 66 - 138) - 11, 7), (1, 0, 0), (18, 138 - 11, 7)]
    w = BitWriter(); dynamic_block(w, True, ll, [0] * 30, [65, 65], clseq=seq, hlit=257, hdist=30)
    out.append(("repeat-past-end", finish(w, b"AA"), 2))
    # 11. fixed block: legal matches incl. length 258 and an overlapping distance 1
    f_ops = [1, 2, 3, 4, ('m', 258, 4), ('m', 10, 1), 200, 255, ('m', 3, 262)]
    w = BitWriter(); fixed_block(w, True, f_ops)
    out.append(("fixed-legal", finish(w, plain_of([f_ops])), len(plain_of([f_ops]))))
    # 12. fixed block using literal/length symbol 286
    w = BitWriter(); fixed_block(w, True, [65, ('sym', 286)])
    out.append(("fixed-sym-286", finish(w, b"A"), 8))
    # 13. fixed block using distance symbol 30
    w = BitWriter(); fixed_block(w, True, [65, 66, 67, ('dsym', 3, 30, 0)])
    out.append(("fixed-dist-30", finish(w, b"ABC"), 8))
    # 14. distance reaching before the start of the output
    w = BitWriter(); fixed_block(w, True, [65, 66, ('m', 3, 3)])
    out.append(("dist-too-far", finish(w, b"AB"), 8))
    # 15. stored block with a bad complement, and a good one with odd alignment behind a fixed block
    w = BitWriter(); stored_block(w, True, b"hello", nlen_xor=0xfffe)
    out.append(("stored-bad-nlen", finish(w, b"hello"), 5))
    w = BitWriter(); fixed_block(w, False, [104]); stored_block(w, False, b"ello "); fixed_block(w, True, [119, ('m', 3, 3)])
    out.append(("fixed-stored-fixed", finish(w, b"hello wo w"), 10))
    # 16. wrong Adler-32, bad header check, preset dictionary flag, window bits 8 (legal), compression method 7
    w = BitWriter(); fixed_block(w, True, [65])
    out.append(("bad-adler", finish(w, b"A", adler=1), 1))
    out.append(("bad-fcheck", finish(w, b"A", flg=0x9d), 1))
    out.append(("fdict", finish(w, b"A", flg=0xbb), 1))
    out.append(("cinfo-0", finish(w, b"A", cmf=0x08, flg=0x1d), 1))
    out.append(("cm-7", finish(w, b"A", cmf=0x77, flg=0x85 + ((31 - (0x7785 % 31)) % 31)), 1))
    # 17. two-symbol distance code of one bit each (complete)
    w = BitWriter(); dynamic_block(w, True, lit_ab_len, [1, 0, 1], [65, 66, 67, ('m', 3, 3), ('m', 3, 1)])
    out.append(("dist-two-symbols", finish(w, plain_of([[65, 66, 67, ('m', 3, 3), ('m', 3, 1)]])), 9))
    # 18. one-symbol distance code, the OTHER (unassigned) bit pattern used
    w = BitWriter(); lit = canonical(lit_ab_len); dynamic_block(w, True, lit_ab_len, [0, 0, 1], [65, 66, 67])
    # re-open: replace the end-of-block by a length symbol followed by distance bit 1
    w2 = BitWriter(); dynamic_block(w2, True, lit_ab_len, [0, 0, 1], [65, 66, 67, ('sym', 257)])
    # ('sym', 257) wrote length 3; now the distance bit and the end-of-block must follow by hand: rebuild explicitly
    w3 = BitWriter()
    seq = cl_sequence(list(lit_ab_len) + [0, 0, 1])
    dynamic_block(w3, True, lit_ab_len, [0, 0, 1], [], clseq=seq)
    out.append(("dist-one-symbol-eob-only", finish(w3, b""), 4))
    # 19. HLIT = 287 symbols (> 286)
    w = BitWriter(); big = [0] * 287; big[65] = 1; big[256] = 1
    dynamic_block(w, True, big, [0], [65])
    out.append(("hlit-287", finish(w, b"A"), 1))
    # 20. fixed block cut short (no end-of-block), and an empty final fixed block
    w = BitWriter(); w.bits(1, 1); w.bits(1, 2); w.code(*canonical([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8)[65])
    out.append(("fixed-no-eob", finish(w, b"A"), 1))
    w = BitWriter(); fixed_block(w, True, [])
    out.append(("fixed-empty", finish(w, b""), 4))
    # 21. block type 3
    w = BitWriter(); w.bits(1, 1); w.bits(3, 2)
    out.append(("btype-3", finish(w, b""), 4))
    return out
