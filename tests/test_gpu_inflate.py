"""vk_inflate_device (gzip inflate in HBM) against zlib / Python's gzip module on the host: the
compressed files step D of the reference really receives (commands/image.py:696-708) must come out
byte for byte."""
import gzip
import io
import struct
import zlib

import numpy as np
import pytest

from varkoder_amd import _capi, synth

pytestmark = pytest.mark.gpu


def gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=31):
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
    return co.compress(data) + co.flush()


def run(eng, files, caps=None):
    import torch
    offs, pos = [], 0
    for f in files:
        offs.append(pos)
        pos += (len(f) + 15) // 16 * 16
    host = np.zeros(pos + 16, dtype=np.uint8)
    for o, f in zip(offs, files):
        host[o:o + len(f)] = np.frombuffer(f, dtype=np.uint8)
    dev = torch.from_numpy(host).cuda()
    if caps is None:
        caps = [struct.unpack("<I", f[-4:])[0] if len(f) >= 4 else 0 for f in files]
    ooffs, pos = [], 0
    for c in caps:
        ooffs.append(pos)
        pos += (c + 15) // 16 * 16
    out = torch.full((pos + 16,), 0xEE, dtype=torch.uint8, device="cuda")
    lens, status = eng.inflate(dev, np.array(offs, dtype=np.uint64), np.array([len(f) for f in files], dtype=np.uint64),
                               out, np.array(ooffs, dtype=np.uint64), np.array(caps, dtype=np.uint64))
    res = out.cpu().numpy()
    return [bytes(res[o:o + int(n)]) for o, n in zip(ooffs, lens)], status, res, ooffs


def test_inflate_matches_zlib_on_every_block_type(engines):
    eng = engines(7)
    rng = np.random.default_rng(5)
    fq = synth.sample_fastq(3, 3000, 150, dist=1).tobytes()
    noise = rng.integers(0, 256, size=70000, dtype=np.uint8).tobytes()
    texts = {
        "fastq_l1": (fq, 1), "fastq_l6": (fq, 6), "fastq_l9": (fq, 9), "stored_l0": (fq[:200000], 0),
        "noise_l6": (noise, 6),                                   # incompressible: stored blocks inside a stream
        "empty": (b"", 6), "one_byte": (b"A", 6), "rle": (b"G" * 100000, 9),
        "period3": (b"ACG" * 40000, 6), "far_matches": (noise[:32768] + noise[:32768] + noise[100:32768], 9),
        "text": (b"".join(b"line %d of some text\n" % i for i in range(20000)), 6),
    }
    names = list(texts)
    files = [gz(*texts[n]) for n in names]
    files.append(gz(fq[:50000], 6, zlib.Z_FIXED))                # fixed Huffman codes only
    names.append("fixed")
    texts["fixed"] = (fq[:50000], 6)
    files.append(gz(fq, 6, zlib.Z_HUFFMAN_ONLY))                 # literals only, long codes
    names.append("huffman_only")
    texts["huffman_only"] = (fq, 6)
    got, status, _, _ = run(eng, files)
    for n, g, st in zip(names, got, status):
        assert st == 0, (n, st)
        assert g == texts[n][0], n


def test_inflate_headers_members_and_padding(engines):
    eng = engines(7)
    a = synth.sample_fastq(1, 500, 150).tobytes()
    b = synth.sample_fastq(2, 700, 100).tobytes()
    buf = io.BytesIO()
    with gzip.GzipFile(filename="sample@00001000K.fq", mode="wb", fileobj=buf, mtime=12345) as f:   # FNAME
        f.write(a)
    named = buf.getvalue()
    # FEXTRA + FCOMMENT + FHCRC written by hand around a raw deflate stream
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = raw.compress(b) + raw.flush()
    hdr = bytes([0x1f, 0x8b, 8, 4 | 16 | 2, 0, 0, 0, 0, 0, 3]) + struct.pack("<H", 6) + b"BC\x02\x00\x07\x00" + b"a comment\x00"
    hdr += struct.pack("<H", zlib.crc32(hdr) & 0xFFFF)
    fancy = hdr + body + struct.pack("<II", zlib.crc32(b), len(b))
    multi = gz(a) + gz(b) + gz(b"") + gz(a[:1000])
    padded = gz(a) + b"\x00" * 37
    files = [named, fancy, multi, padded]
    want = [a, b, a + b + a[:1000], a]
    got, status, _, _ = run(eng, files, caps=[len(w) for w in want])
    assert status.tolist() == [0, 0, 0, 0]
    assert got == want
    assert gzip.decompress(multi) == want[2] and gzip.decompress(padded) == a       # what the host tools make of them


def test_inflate_flags_bad_streams_and_never_writes_past_its_slot(engines):
    eng = engines(7)
    a = synth.sample_fastq(4, 2000, 150).tobytes()
    good = gz(a)
    cut = good[: len(good) // 2]
    flipped = bytearray(good)
    for i in range(200, len(flipped) - 8, 97):
        flipped[i] ^= 0x5A
    notgz = b"@r1\nACGT\n+\nIIII\n" * 10
    wrong_size = good[:-4] + struct.pack("<I", len(a) + 1)
    files = [good, cut, bytes(flipped), notgz, wrong_size, good]
    caps = [len(a), len(a), len(a), 64, len(a) + 16, len(a) // 3]          # the last one is too small on purpose
    got, status, res, ooffs = run(eng, files, caps=caps)
    assert status[0] == 0 and got[0] == a
    assert status[1] & (_capi.VK_GZ_TRUNCATED | _capi.VK_GZ_BAD_DATA)
    assert status[2] != 0
    assert status[3] & _capi.VK_GZ_BAD_HEADER
    assert status[4] & _capi.VK_GZ_BAD_SIZE
    assert status[5] & _capi.VK_GZ_OVERFLOW
    # nothing outside a file's own slot was touched: the filler between and after the slots is intact
    for o, c, nxt in zip(ooffs, caps, ooffs[1:] + [res.size - 16]):
        pad_from = o + c
        assert (res[pad_from:nxt] == 0xEE).all()
    assert (res[-16:] == 0xEE).all()


def test_inflate_many_files_and_a_big_one(engines):
    eng = engines(7)
    big = synth.sample_fastq(9, 200000, 150).tobytes()            # 64 MB of text
    files = [gz(big, 6)] + [gz(synth.sample_fastq(100 + i, 300 + 17 * i, 150, dist=i & 1).tobytes(), 1 + i % 9) for i in range(40)]
    got, status, _, _ = run(eng, files)
    assert not status.any()
    assert got[0] == big
    for i in range(40):
        assert got[1 + i] == synth.sample_fastq(100 + i, 300 + 17 * i, 150, dist=i & 1).tobytes(), i


def test_inflate_large_files_take_many_wavefronts_and_agree(engines):
    """Files of 512 KiB and more go through the chunked path (block starts found in every 256 KiB of the
    compressed stream, chunks decoded with a symbolic window, windows propagated, markers resolved):
    same bytes as zlib for every level, for streams with stored blocks and long runs in them, and for a
    large multi-member file."""
    eng = engines(7)
    rng = np.random.default_rng(11)
    fq = [synth.sample_fastq(20 + i, 60000, 150, dist=i & 1).tobytes() for i in range(3)]      # 19 MB each
    noise = rng.integers(0, 256, size=3_000_000, dtype=np.uint8).tobytes()
    mixed = fq[0][:4_000_000] + noise + b"A" * 2_000_000 + fq[1][:5_000_000] + noise[:700_000] + b"ACGT" * 500_000
    texts = [fq[0], fq[1], fq[2], mixed, fq[0] + fq[1]]
    files = [gz(fq[0], 1), gz(fq[1], 6), gz(fq[2], 9), gz(mixed, 6), gz(fq[0], 6) + gz(fq[1], 4)]
    assert all(len(f) >= 2 * (1 << 18) for f in files)
    got, status, _, _ = run(eng, files, caps=[len(t) for t in texts])
    assert status.tolist() == [0] * len(files)
    for i, (g, t) in enumerate(zip(got, texts)):
        assert len(g) == len(t), i
        assert g == t, i
    # a multi-member file whose caller knows only the last member's size: told to come back with more room
    got, status, _, _ = run(eng, [files[4]])
    assert status[0] == _capi.VK_GZ_OVERFLOW
    # damage in the middle of a large file is reported, not decoded around
    bad = bytearray(files[1])
    bad[len(bad) // 2] ^= 0xFF
    bad[len(bad) // 2 + 1] ^= 0xFF
    got, status, _, _ = run(eng, [bytes(bad)], caps=[len(texts[1])])
    assert status[0] != 0 or got[0] != texts[1]
    assert status[0] != 0


def test_engine_uploads_gzip_files_and_counts_them(engines, tmp_path):
    """ImageEngine.upload_files: plain and gzip files side by side, a multi-member file whose last size
    word promises less than it holds (inflated again into a side buffer), and a damaged file that ends
    up as an empty sample; the histograms of the rest equal the oracle's."""
    from oracle import oracle
    eng = engines(7)
    a = synth.sample_fastq(61, 3000, 150).tobytes()
    b = synth.sample_fastq(62, 1200, 150, dist=1).tobytes()
    (tmp_path / "plain.fq").write_bytes(a)
    (tmp_path / "one.fq.gz").write_bytes(gz(b, 6))
    (tmp_path / "multi.fq.gz").write_bytes(gz(a, 1) + gz(b, 9))          # last ISIZE = len(b) < len(a + b)
    bad = bytearray(gz(a, 6))
    for i in range(300, len(bad) - 8, 211):
        bad[i] ^= 0x33
    (tmp_path / "bad.fq.gz").write_bytes(bytes(bad))
    paths = [tmp_path / "one.fq.gz", tmp_path / "plain.fq", tmp_path / "bad.fq.gz", tmp_path / "multi.fq.gz"]
    dev, offs, lens = eng.upload_files(paths)
    host = dev.cpu().numpy()
    want = [b, a, b"", a + b]
    for o, n, w in zip(offs, lens, want):
        assert int(n) == len(w)
        assert bytes(host[int(o):int(o) + int(n)]) == w
    hist, status = eng.count(dev, offs, lens)
    got = hist.cpu().numpy().view(np.uint32)
    for i, w in enumerate(want):
        assert np.array_equal(got[i], oracle.count_fastq(w, 7)[0]), i


def test_inflate_fuzz_small_streams(engines):
    """A few hundred streams from zlib with random level / strategy / window / memLevel over random content
    kinds, plus streams cut into many flush points (many small blocks, empty stored blocks): all byte-exact."""
    eng = engines(7)
    rng = np.random.default_rng(2025)
    fq = synth.sample_fastq(77, 2000, 150, dist=1).tobytes()
    files, texts = [], []
    for i in range(300):
        kind = int(rng.integers(0, 6))
        n = int(rng.integers(0, 120_000)) if i % 7 else int(rng.integers(0, 40))
        if kind == 0:
            t = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        elif kind == 1:
            t = rng.integers(65, 70, size=n, dtype=np.uint8).tobytes()
        elif kind == 2:
            o = int(rng.integers(0, max(1, len(fq) - n)))
            t = fq[o:o + n]
        elif kind == 3:
            unit = rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
            t = (unit * (n // len(unit) + 1))[:n]
        elif kind == 4:
            t = bytes(rng.integers(0, 4, size=n, dtype=np.uint8) * 17 + 33)
        else:
            t = b"".join(b"%d\t%x\n" % (j * 7919 % 100003, j) for j in range(n // 12))
        co = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, 16 + int(rng.integers(9, 16)), int(rng.integers(1, 10)),
                              int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])))
        if i % 5 == 0 and len(t) > 100:      # many blocks: flushes every few hundred bytes
            z, step = b"", max(50, len(t) // 40)
            for o in range(0, len(t), step):
                z += co.compress(t[o:o + step]) + co.flush(zlib.Z_FULL_FLUSH if (o // step) % 3 == 0 else zlib.Z_SYNC_FLUSH)
            z += co.flush()
        else:
            z = co.compress(t) + co.flush()
        files.append(z)
        texts.append(t)
    got, status, _, _ = run(eng, files, caps=[len(t) for t in texts])
    assert not status.any(), np.flatnonzero(status)
    for i, (g, t) in enumerate(zip(got, texts)):
        assert g == t, i


def test_inflate_large_files_without_usable_block_starts(engines):
    """Large files the chunked path cannot split: stored blocks only (no dynamic-codes header anywhere), and a
    gzip file stored inside another one -- every block header of the inner file sits, byte-aligned, in the
    outer file's stored blocks, so the finder reports starts that the chunk before never reaches; the decoder
    must step over them (or hand the file to the one-wavefront kernel)."""
    eng = engines(7)
    fq = synth.sample_fastq(88, 30000, 150).tobytes()                  # 9.6 MB
    inner = gz(fq, 6)                                                  # ~1.6 MB, dozens of dynamic blocks
    stored_big = gz(fq[:3_000_000], 0)
    nested = gz(inner, 0)
    assert len(stored_big) >= 2 * (1 << 18) and len(nested) >= 2 * (1 << 18)
    got, status, _, _ = run(eng, [stored_big, nested], caps=[3_000_000, len(inner)])
    assert status.tolist() == [0, 0]
    assert got[0] == fq[:3_000_000] and got[1] == inner


def test_inflate_verifies_the_crc_of_single_member_files(engines):
    """The trailer's CRC-32 is recomputed on the GPU from the inflated text (segments, GF(2) shift operators)
    for every file that consists of one member, on the one-wavefront path and on the chunked path alike: a
    wrong check word, or damage that leaves the stream's structure and size intact (a changed byte inside a
    stored block), sets VK_GZ_BAD_CRC and nothing else."""
    eng = engines(7)
    small = synth.sample_fastq(31, 3000, 150).tobytes()
    large = synth.sample_fastq(32, 60000, 150, dist=1).tobytes()           # 19 MB: chunked path at level 6
    texts = [small, large, small[:70001], b"", b"A", b"\0" * 100000, large[: (1 << 16) * 5], large[: (1 << 16) * 5 + 1],
             small[:1023], small[:1024], small[:1025]]
    files = [gz(t, 1 + i % 9) for i, t in enumerate(texts)]
    assert len(files[1]) >= 2 * (1 << 18)
    got, status, _, _ = run(eng, files)
    assert status.tolist() == [0] * len(files)
    assert all(g == t for g, t in zip(got, texts))

    def with_crc(f, delta):
        word = struct.unpack("<I", f[-8:-4])[0]
        return f[:-8] + struct.pack("<I", (word + delta) & 0xFFFFFFFF) + f[-4:]

    bad = [with_crc(f, 1 + i) for i, f in enumerate(files)]
    got, status, _, _ = run(eng, bad)
    assert status.tolist() == [_capi.VK_GZ_BAD_CRC] * len(bad)
    assert all(g == t for g, t in zip(got, texts))                         # the text itself is still delivered

    # a changed byte inside stored blocks: only the check word can tell
    for text in (small, large):
        f = bytearray(gz(text, 0))
        assert len(f) > len(text)
        at = len(f) // 2
        f[at] ^= 0x01
        want = bytearray(text)
        got, status, _, _ = run(eng, [bytes(f)])
        assert status[0] == _capi.VK_GZ_BAD_CRC, status
        assert len(got[0]) == len(want) and got[0] != text
    # several members: every member's check word is verified against its own stretch of the text
    two_bad = with_crc(gz(small), 5) + gz(small[:1000])
    two_good = gz(small) + gz(b"") + gz(small[:1000])
    got, status, _, _ = run(eng, [two_bad, two_good], caps=[len(small) + 1000] * 2)
    assert status.tolist() == [_capi.VK_GZ_BAD_CRC, 0] and got[0] == got[1] == small + small[:1000]


def test_inflate_chunk_sizes_and_false_block_starts(monkeypatch):
    """The chunked path with the chunk size forced (VKIMG_GZ_CHUNK_BYTES, read when a context is made) from 64 KiB
    -- smaller than some DEFLATE blocks, so chunks without any block start occur -- to 1 MiB: same text every
    time.  The input carries, inside stored blocks, bytes that look like the start of a dynamic-codes block (the
    header of a real one, copied): the finder must not settle on them (it decodes a candidate block to its end
    and wants another block header there), and whatever it settles on, the text must come out right."""
    from varkoder_amd.engine import ImageEngine
    rng = np.random.default_rng(23)
    fq = synth.sample_fastq(40, 60000, 150, dist=1).tobytes()                       # 19 MB
    real = gz(fq[:3_000_000], 6, wbits=-15)                                          # raw deflate: begins with a dynamic header
    decoy = real[:4096]
    noise = rng.integers(0, 256, size=1_500_000, dtype=np.uint8).tobytes()
    spiked = b"".join(noise[i:i + 50_000] + decoy for i in range(0, len(noise), 50_000))
    texts = [fq, fq[:5_000_000] + spiked + fq[5_000_000:9_000_000] + spiked[:300_000] + fq[9_000_000:]]
    files = [gz(texts[0], 6), gz(texts[1], 6)]
    assert all(len(f) >= (1 << 19) for f in files)
    for cb in (1 << 16, 1 << 17, 3 << 16, 1 << 20):
        monkeypatch.setenv("VKIMG_GZ_CHUNK_BYTES", str(cb))
        eng = ImageEngine(k=7, mapping="cgr", device=0)
        try:
            got, status, _, _ = run(eng, files, caps=[len(t) for t in texts])
        finally:
            eng.close()
        assert status.tolist() == [0, 0], cb
        assert got[0] == texts[0] and got[1] == texts[1], cb


def bgzf(data, block=30000, level=6):
    """A BGZF file (bgzip; what BBTools writes through bgzip): members of at most 64 KiB with a 'BC' extra
    field that holds the member's size, and an empty member at the end."""
    out = []
    for i in list(range(0, len(data), block)) + [None]:
        chunk = data[i:i + block] if i is not None else b""
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        raw = co.compress(chunk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(raw) + 25) + raw +
                   struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return b"".join(out)


def test_many_member_files_are_sized_up_front_and_inflated_once(engines, tmp_path):
    """A BGZF file of more than 1000 members ends with an empty member (ISIZE 0): the text slot comes from a
    walk over the block headers (engine.bgzf_text_size), not from the file's last four bytes, so the file is
    inflated ONCE; a plain concatenation of members (no size fields to walk) is inflated a second time into
    a slot of the size the first pass reported; bytes after the last member that are not gzip are ignored, as
    zlib's gzread (dsk) and gzip(1) ignore them."""
    from oracle import oracle
    from varkoder_amd import engine as engine_mod
    eng = engines(7)
    big = synth.sample_fastq(11, 110000, 150, dist=1).tobytes()          # 35 MB: 1174 members of 30000 B
    small = synth.sample_fastq(12, 3000, 150).tobytes()
    files = {"bgzf_big.fq.gz": bgzf(big), "bgzf_small.fq.gz": bgzf(small, block=65280),
             "concat.fq.gz": b"".join(gz(big[i:i + 3_000_000], 1) for i in range(0, len(big), 3_000_000)),
             "rle_members.fq.gz": gz(b"@r\n" + b"A" * 4_000_000 + b"\n+\n" + b"I" * 4_000_000 + b"\n", 9) + gz(small),
             "garbage_tail.fq.gz": gz(small) + b"\x00\x00not a gzip member at all" * 3,
             "plain_single.fq.gz": gz(small, 1)}
    assert len(bgzf(big)) > 0 and engine_mod.bgzf_text_size(files["bgzf_big.fq.gz"]) == len(big)
    assert files["bgzf_big.fq.gz"].count(b"\x1f\x8b\x08\x04") >= 1000
    want = {"bgzf_big.fq.gz": big, "bgzf_small.fq.gz": small, "concat.fq.gz": big,
            "rle_members.fq.gz": b"@r\n" + b"A" * 4_000_000 + b"\n+\n" + b"I" * 4_000_000 + b"\n" + small,
            "garbage_tail.fq.gz": small, "plain_single.fq.gz": small}
    for name, blob in files.items():
        (tmp_path / name).write_bytes(blob)
        if name != "garbage_tail.fq.gz":
            assert gzip.decompress(blob) == want[name]
    calls = []
    real = eng.inflate

    def counting(gzbuf, go, gl, out, oo, oc):
        calls.append(len(go))
        return real(gzbuf, go, gl, out, oo, oc)
    eng.inflate = counting
    try:
        for names in (["bgzf_big.fq.gz", "bgzf_small.fq.gz", "plain_single.fq.gz"], ["concat.fq.gz"],
                      ["rle_members.fq.gz", "garbage_tail.fq.gz"]):
            calls.clear()
            dev, offs, lens = eng.upload_files([tmp_path / n for n in names])
            text = dev.cpu().numpy()
            for n, o, ln in zip(names, offs, lens):
                assert bytes(text[int(o):int(o) + int(ln)]) == want[n], n
            if names[0].startswith("bgzf"):
                assert len(calls) == 1 and calls[0] > 1000, calls   # ONE inflate call (a job per BGZF member), nothing inflated twice
            elif names[0] == "concat.fq.gz":
                assert calls == [1, 1], calls           # once more, into a slot of the size the first pass reported
            else:
                assert calls == [2, 1, 1], calls        # 35:1 over a small file: the 32x slot overflows too, 256x holds it
            hist, status = eng.count(dev, offs, lens)
            for i, n in enumerate(names):
                wh, _, st = oracle.count_fastq(want[n], 7)
                assert st == 0 and int(status.cpu()[i]) == 0
                assert np.array_equal(hist.cpu().numpy().view(np.uint32)[i], wh), n
    finally:
        eng.inflate = real


def test_a_truncated_gzip_file_cannot_reserve_gigabytes(engines, tmp_path):
    """The last four bytes of a truncated .fq.gz are arbitrary: the text slot is capped at 64x the file's size,
    the file comes out empty with a status, its batch mates are untouched."""
    eng = engines(7)
    a = synth.sample_fastq(5, 3000, 150).tobytes()
    good = gz(a)
    cut = good[: len(good) * 2 // 3 - 4] + b"\xff\xff\xff\xf0"           # "ISIZE" = 4 GiB - 256 MiB
    (tmp_path / "good.fq.gz").write_bytes(good)
    (tmp_path / "cut.fq.gz").write_bytes(cut)
    st = eng.stage_files([tmp_path / "cut.fq.gz", tmp_path / "good.fq.gz"])
    assert int(st["caps"][0]) <= 64 * len(cut) + (1 << 16) and int(st["caps"][1]) == len(a)
    dev, offs, lens = eng.upload_staged(st)
    assert int(lens[0]) == 0 and int(lens[1]) == len(a)
    assert bytes(dev.cpu().numpy()[int(offs[1]):int(offs[1]) + len(a)]) == a


def test_inflate_verifies_every_member_of_many_member_files(engines, tmp_path):
    """BGZF and concatenated files: a wrong check word in ONE member in the middle -- sizes all right, text all
    right -- is VK_GZ_BAD_CRC.  Through the C ABI (a small BGZF file on the one-wavefront path, a large
    concatenation on the chunked path with members crossing chunk borders) and through the file route, where a
    BGZF file is inflated member by member."""
    eng = engines(7)
    small = synth.sample_fastq(41, 2500, 150).tobytes()                     # 0.8 MB of text
    large = synth.sample_fastq(42, 120000, 150, dist=1).tobytes()           # 38 MB

    def flip_member_crc(blob, which):
        """The CRC-32 word of member `which` (0-based) of a BGZF file, plus one."""
        pos = 0
        for _ in range(which):
            pos += (blob[pos + 16] | (blob[pos + 17] << 8)) + 1
        end = pos + (blob[pos + 16] | (blob[pos + 17] << 8)) + 1
        word = struct.unpack("<I", blob[end - 8:end - 4])[0]
        return blob[:end - 8] + struct.pack("<I", (word + 1) & 0xFFFFFFFF) + blob[end - 4:]

    def members(blob):
        return blob.count(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC")

    good = bgzf(small, block=20000)
    nm = members(good)
    assert 30 < nm <= 62
    files = [good, flip_member_crc(good, nm // 2), flip_member_crc(good, nm - 1)]      # (the last one: the empty end marker)
    texts, bad = [small] * 3, [False, True, True]
    cat = [gz(large[i:i + 2_500_000], 1 + (i // 2_500_000) % 9) for i in range(0, len(large), 2_500_000)]
    files.append(b"".join(cat))
    j = len(cat) // 2
    word = struct.unpack("<I", cat[j][-8:-4])[0]
    cat[j] = cat[j][:-8] + struct.pack("<I", word ^ 0x10) + cat[j][-4:]
    files.append(b"".join(cat))
    texts += [large, large]
    bad += [False, True]
    got, status, _, _ = run(eng, files, caps=[len(t) for t in texts])
    assert [bool(x) for x in status.tolist()] == bad, status.tolist()
    assert all(x in (0, _capi.VK_GZ_BAD_CRC) for x in status.tolist())
    assert all(g == t for g, t in zip(got, texts))
    # the file route: thousands of members, each inflated and checked as a file of its own
    for block in (30000, 65280):
        blob = bgzf(large, block=block)
        n = members(blob)
        names = []
        for tag, data in (("ok", blob), ("mid", flip_member_crc(blob, n // 2)), ("end", flip_member_crc(blob, n - 1))):
            names.append(tmp_path / f"{tag}_{block}.fq.gz")
            names[-1].write_bytes(data)
        dev, offs, lens = eng.upload_files(names)
        assert [int(x) for x in lens] == [len(large), 0, 0]
        assert bytes(dev.cpu().numpy()[int(offs[0]):int(offs[0]) + len(large)]) == large


def test_overflow_retries_never_shrink_the_slot_and_mapped_files_are_released(engines, tmp_path, monkeypatch):
    """A small, highly compressible file whose size word lies (two members: the slot is sized for the last one
    alone) overflows its first slot; the retry must be larger than that slot (32x the file was below
    the 64x first slot), and the text must come out whole.  Also through the mapped route, whose pinned pages are
    released whatever happens."""
    from varkoder_amd import engine as E
    eng = engines(7)
    a = (b"@r\n" + b"A" * 150 + b"\n+\n" + b"I" * 150 + b"\n") * 4000        # 1.2 MB of text, compresses ~200x
    b = b"@q\nACGTACGTAC\n+\nIIIIIIIIII\n"
    blob = gz(a, 9) + gz(b, 9)                                                # ISIZE of the file = len(b)
    f = tmp_path / "two.fq.gz"
    f.write_bytes(blob)
    for route in ("0", "1"):
        monkeypatch.setattr(E, "USE_MAPPED_UPLOAD", route == "1")
        dev, offs, lens = eng.upload_files([f])
        assert int(lens[0]) == len(a) + len(b), route
        assert bytes(dev.cpu().numpy()[int(offs[0]):int(offs[0]) + int(lens[0])]) == a + b, route
        assert int(eng.last_upload_status[0]) == 0


def test_sources_just_behind_the_history_across_block_kinds(engines):
    """The slab resolver takes a source at most ~448 positions back from its LDS history and anything further from the text
    in memory, four slabs' loads at once behind a counted wait for this wavefront's older stores (csrc/vk_inflate.h,
    gz_resolve_slabs: vmcnt(7)).  Streams made for the seam: every unit repeats text 513..600 positions back (just behind
    the 512-element history), the next one text under 448 back, on and on -- through dynamic blocks, stored blocks (which go
    around the history) and full flushes in between, and with the short token groups that flushes end in; small files (one
    wavefront, text bytes) and large ones (chunk decoder, u16 elements with the unknown window).  Byte-exact against zlib,
    every member's CRC checked on the device."""
    eng = engines(7)
    rng = np.random.default_rng(606)

    def seam_text(n, seed):
        r = np.random.default_rng(seed)
        out = bytearray(r.integers(65, 91, size=700, dtype=np.uint8).tobytes())
        far = True
        while len(out) < n:
            d = int(r.integers(513, 601)) if far else int(r.integers(3, 449))
            ln = int(r.integers(4, 40))
            start = len(out) - d
            for i in range(ln):            # (byte by byte: an overlapping copy when ln > d)
                out.append(out[start + i])
            out.extend(r.integers(65, 91, size=int(r.integers(0, 3)), dtype=np.uint8).tobytes())   # a literal or two between
            far = not far
        return bytes(out[:n])

    def mixed_stream(text, level, seed):
        """one gzip member whose blocks alternate: dynamic (or fixed) / stored / sync- and full-flushed pieces"""
        r = np.random.default_rng(seed)
        co = zlib.compressobj(level, zlib.DEFLATED, 31, 9)
        out, pos = [], 0
        while pos < len(text):
            n = int(r.integers(900, 9000))
            piece = text[pos:pos + n]
            pos += n
            kind = int(r.integers(0, 4))
            out.append(co.compress(piece))
            if kind == 0:
                out.append(co.flush(zlib.Z_SYNC_FLUSH))     # ends the block, an empty stored block behind it
            elif kind == 1:
                out.append(co.flush(zlib.Z_FULL_FLUSH))     # the same, and the next block may not look back
            elif kind == 2:                                 # a real stored block: incompressible bytes in between
                noise = r.integers(0, 256, size=int(r.integers(100, 700)), dtype=np.uint8).tobytes()
                out.append(co.compress(noise))
                out.append(co.flush(zlib.Z_SYNC_FLUSH))
                text = text[:pos] + noise + text[pos:]
                pos += len(noise)
        out.append(co.flush())
        return b"".join(out), text

    files, wants = [], []
    for i, (n, level) in enumerate([(60_000, 1), (60_000, 6), (200_000, 1), (200_000, 9), (3_000_000, 1), (3_000_000, 6)]):
        f, t = mixed_stream(seam_text(n, 100 + i), level, 200 + i)
        assert zlib.decompress(f, 31) == t
        files.append(f)
        wants.append(t)
    got, status, _, _ = run(eng, files)
    assert [int(x) for x in status] == [0] * len(files)
    for g, w in zip(got, wants):
        assert g == w
