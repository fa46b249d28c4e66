"""CPU model of k_jpeg_huff's state-only decoder and its synchronisation rounds (not part of the library): the scan's bits in
one big integer, one 16-bit lookup per Huffman table giving (bits consumed, coefficients advanced) per symbol.  A segment
decoded from a state (bit position, block within the MCU, coefficient index) leaves in a state, exactly as
jpeg_state_segment does (k_jpeg.hip); `simulate` reproduces the kernel's rounds (tools/jpeg_rounds.py prints the same counts).
Round 3 had a bit-by-bit version of this model (removed with the round-4 pruning); this one is ~30x faster."""
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FILES = sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'sample-images1', '*.jpg')))


def parse(data):
    """-> DHT tables {(class, id): (counts, symbols)}, frame components, scan components, the entropy-coded bytes (unstuffed)"""
    i = 2
    dht = {}
    comps = sel = None
    while i < len(data):
        m = data[i + 1]
        i += 2
        if m == 0xD8 or 0xD0 <= m <= 0xD7:
            continue
        ln = (data[i] << 8) | data[i + 1]
        seg = data[i + 2:i + ln]
        if m == 0xC4:
            o = 0
            while o < len(seg):
                counts = list(seg[o + 1:o + 17])
                tot = sum(counts)
                dht[(seg[o] >> 4, seg[o] & 15)] = (counts, list(seg[o + 17:o + 17 + tot]))
                o += 17 + tot
        elif m == 0xC0:
            comps = [(seg[6 + 3 * k], seg[7 + 3 * k] >> 4, seg[7 + 3 * k] & 15) for k in range(seg[5])]
        elif m == 0xDA:
            sel = [(seg[1 + 2 * k], seg[2 + 2 * k] >> 4, seg[2 + 2 * k] & 15) for k in range(seg[0])]
            i += ln
            break
        i += ln
    scan = bytearray()
    while i < len(data):
        b = data[i]
        if b == 0xFF:
            n = data[i + 1]
            if n == 0:
                scan.append(0xFF)
            elif not 0xD0 <= n <= 0xD7:
                break
            i += 2
            continue
        scan.append(b)
        i += 1
    return dht, comps, sel, bytes(scan)


class Dec:
    def __init__(self, data):
        dht, comps, sel, scan = parse(data)
        self.nbits = len(scan) * 8
        self.big = int.from_bytes(scan + b'\xff' * 16, 'big')
        self.total = self.nbits + 128
        layout = []
        for (cid, h, v), (sid, td, ta) in zip(comps, sel):
            layout += [(td, ta)] * (h * v if len(comps) > 1 else 1)
        self.bpm = len(layout)
        lut = {}
        for (tc, th), (counts, vals) in dht.items():
            tab = [(16, 1 if tc == 0 else 64)] * 65536   # no code at all: a 16-bit zero DC difference / end of block (as the kernel)
            code = k = 0
            for ln in range(1, 17):
                for _ in range(counts[ln - 1]):
                    s = vals[k]
                    k += 1
                    ent = (ln + s, 1) if tc == 0 else (ln + (s & 15), (s >> 4) + 1 if s & 15 else (16 if s >> 4 == 15 else 64))
                    lo = code << (16 - ln)
                    tab[lo:lo + (1 << (16 - ln))] = [ent] * (1 << (16 - ln))
                    code += 1
                code <<= 1
            lut[(tc, th)] = tab
        self.dcl = [lut[(0, td)] for td, ta in layout]
        self.acl = [lut[(1, ta)] for td, ta in layout]

    def run(self, p, blk, k, p_end):
        """one segment from state (p, blk, k): -> exit state, blocks completed"""
        nblk = 0
        big, tot, bpm = self.big, self.total, self.bpm
        while p < p_end:
            w = (big >> (tot - p - 16)) & 0xffff
            if k == 0:
                p += self.dcl[blk][w][0]
                k = 1
            else:
                n, a = self.acl[blk][w]
                p += n
                k += a
            if k >= 64:
                k = 0
                nblk += 1
                blk = blk + 1 if blk + 1 < bpm else 0
        return (p, blk, k), nblk

    def steps(self, tab_bits, nsym):
        """decode steps of the whole scan with a first-level table of tab_bits bits holding up to nsym symbols per entry
        (the kernel: 10 bits, 2 symbols, the second one only inside the same block) -> steps, steps that met a long code"""
        p = blk = k = n = nlong = 0
        big, tot, bpm = self.big, self.total, self.bpm
        while p < self.nbits:
            n += 1
            used = cnt = 0
            while cnt < nsym and p < self.nbits:
                w = (big >> (tot - p - 16)) & 0xffff
                nb, a = self.dcl[blk][w] if k == 0 else self.acl[blk][w]
                if cnt == 0 and nb > tab_bits:
                    nlong += 1
                elif used + nb > tab_bits:
                    break
                p += nb
                used += nb
                cnt += 1
                k = 1 if k == 0 else k + a
                if k >= 64:
                    k = 0
                    blk = (blk + 1) % bpm
                    break
                if nb > tab_bits:
                    break
        return n, nlong


def segments(dec, T=512):
    S = 32 * (max(8, (dec.nbits + 32 * T - 1) // (32 * T)) | 1)
    return S, (dec.nbits + S - 1) // S


def truth(dec, T=512):
    S, nseg = segments(dec, T)
    st = (0, 0, 0)
    ex = []
    for i in range(nseg):
        st, _ = dec.run(*st, min((i + 1) * S, dec.nbits + 32))
        ex.append(st)
    return ex


def simulate(dec, T=512, guess=None):
    """the kernel's rounds: -> rounds after the first pass, segments decoded again per round, final exit states"""
    S, nseg = segments(dec, T)
    end = lambda i: min((i + 1) * S, dec.nbits + 32)
    entry = [(i * S, guess(i) if guess else 0, 0) for i in range(nseg)]
    ex = [dec.run(*entry[i], end(i))[0] for i in range(nseg)]
    work = []
    while True:
        ch = [i for i in range(1, nseg) if ex[i - 1] != entry[i]]
        if not ch:
            return len(work), work, ex
        work.append(len(ch))
        new = list(ex)
        for i in ch:
            entry[i] = ex[i - 1]
            new[i] = dec.run(*entry[i], end(i))[0]
        ex = new
