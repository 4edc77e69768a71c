"""Device-resident batch interface to the HIP kernels (one process per GPU).

PyTorch is used for device memory and the stream only; every computation is a call
through the C ABI of include/vkimg.h.  There is no CPU fallback.
"""
import ctypes as C
import mmap
import os
import sys

import numpy as np

from . import _capi
from .config import KMER_MAX, KMER_MIN
from .mapping import pixel_lut, side


def _torch():
    import torch
    return torch


def bgzf_members(buf):
    """The members of a BGZF file (what `bgzip` and BBTools-through-bgzip write: gzip members of at most
    64 KiB, each with its own size in a 'BC' extra field), found by hopping from block header to block header
    -- no inflating: (offsets, compressed sizes, text sizes) as uint64 arrays, or None when `buf` is not BGZF
    from end to end.  The last member of such a file is an empty one (ISIZE 0), so the size word that ends
    the FILE says nothing about the text; and since no member refers to another, every member can be inflated
    by a wavefront of its own."""
    n = len(buf)
    if n == 0:
        return None
    mv = memoryview(buf)
    offs, sizes, texts = [], [], []
    pos = 0
    while pos < n:
        if pos + 18 > n:
            return None
        h = bytes(mv[pos:pos + 18])
        if h[0] != 0x1F or h[1] != 0x8B or h[2] != 8 or not (h[3] & 4) or h[10] != 6 or h[11] != 0 or h[12:16] != b"BC\x02\x00":
            return None
        size = (h[16] | (h[17] << 8)) + 1
        if size < 26 or pos + size > n:
            return None
        text = int.from_bytes(bytes(mv[pos + size - 4:pos + size]), "little")
        if text > 65536:     # a BGZF block holds at most 64 KiB of text: anything else is not BGZF (or hostile) and
            return None      # goes the ordinary way, whose text slot is clamped against the file's size on disk
        offs.append(pos)
        sizes.append(size)
        texts.append(text)
        pos += size
    return (np.array(offs, dtype=np.uint64), np.array(sizes, dtype=np.uint64), np.array(texts, dtype=np.uint64))


def bgzf_text_size(buf):
    """Text bytes of a BGZF file (the sum of its members' ISIZE words), None when `buf` is not BGZF."""
    m = bgzf_members(buf)
    return None if m is None else int(m[2].sum())


# A gzip file's text slot in HBM is reserved before anything is inflated.  The size word that ends the file
# (ISIZE) is the size of its LAST member modulo 2^32 -- right for what split_fastq writes, arbitrary bytes
# in a truncated file: never reserve more than this many times the file's size on disk (+ slack); a file
# that really expands further overflows its slot and is inflated again into one of the size the first pass
# reported (vk_inflate_device, VK_GZ_OVERFLOW).
GZ_MAX_FIRST_RATIO = 64
# How plain-text files reach the GPU: mapped and copied by DMA straight out of the page cache (stage_files /
# vk_upload_mapped: next to no host work) or read() into the pinned staging buffer by the I/O threads (a core per
# ~3 GB/s).  With 16 threads the second is 3-7 % faster on one GPU (both run at the link's rate; measured, DESIGN.md
# 5); a rank that has fewer -- several ranks sharing a host's cores -- cannot feed its link that way.
# VARKODER_AMD_MMAP=1 / 0 forces one or the other; unset: mapped when the rank has fewer than MAPPED_BELOW_THREADS.
USE_MAPPED_UPLOAD = {"1": True, "0": False}.get(os.environ.get("VARKODER_AMD_MMAP", ""))
MAPPED_BELOW_THREADS = 16


def plain_route(threads, engine=None):
    """"mapped" or "staged": how stage_files brings plain-text files in for a rank with this many I/O threads.
    VARKODER_AMD_MMAP decides when set; else what the engine's pipeline has switched to (ImageEngine.route_override:
    pipeline.fastqs_to_images goes over to the mapped route when its staging threads -- read() into pinned memory -- keep
    the copy engine waiting); else mapped below MAPPED_BELOW_THREADS threads."""
    if USE_MAPPED_UPLOAD is not None:
        return "mapped" if USE_MAPPED_UPLOAD else "staged"
    if engine is not None and getattr(engine, "route_override", None) in ("mapped", "staged"):
        return engine.route_override
    return "mapped" if threads < MAPPED_BELOW_THREADS else "staged"


class ImageEngine:
    """FASTQ (in HBM) -> forward k-mer histograms -> uint8 images, for one k and mapping.

    Mirrors steps D+E of run_clean2img (varKoder/commands/image.py:1054-1127) for a
    batch of samples that are already resident on the GPU.
    """

    def __init__(self, k=7, mapping="cgr", device=0, lut=None, npix=None):
        if k not in range(KMER_MIN, KMER_MAX + 1):
            raise ValueError("kmer size must be between 5 and 9")
        torch = _torch()
        if not torch.cuda.is_available():
            raise _capi.VkError(_capi.VK_EHIP, "no GPU visible: the HIP path cannot run (no CPU fallback)")
        self.k = k
        self.mapping = mapping
        self.route_override = None   # see plain_route
        self.device = torch.device("cuda", device)
        self.L = _capi.lib()
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.current_stream(self.device)
        ctx = C.c_void_p()
        _capi.check(None, self.L.vk_ctx_create(device, C.c_void_p(self.stream.cuda_stream), 0, C.byref(ctx)),
                    "vk_ctx_create")
        self.ctx = ctx
        if lut is None:
            self.side = side(k, mapping)
            self.npix = self.side * self.side
            if mapping == "cgr":
                st = self.L.vk_set_mapping(self.ctx, k, None, self.npix)
            else:
                lut = np.ascontiguousarray(pixel_lut(k, mapping), dtype=np.uint32)
                st = self.L.vk_set_mapping(self.ctx, k, lut.ctypes.data_as(C.POINTER(C.c_uint32)), self.npix)
        else:
            lut = np.ascontiguousarray(lut, dtype=np.uint32)
            self.npix = int(npix)
            self.side = int(round(self.npix ** 0.5))
            st = self.L.vk_set_mapping(self.ctx, k, lut.ctypes.data_as(C.POINTER(C.c_uint32)), self.npix)
        _capi.check(self.ctx, st, "vk_set_mapping")
        self.ncode = 4 ** k

    def close(self):
        ex = self.__dict__.pop("_releaser", None)
        if ex is not None:
            ex.shutdown(wait=True)   # mapped files still being unpinned
        if getattr(self, "ctx", None):
            self.L.vk_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------
    @staticmethod
    def _desc(offsets, lengths):
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        lens = np.ascontiguousarray(lengths, dtype=np.uint64)
        if offs.shape != lens.shape or offs.ndim != 1:
            raise ValueError("offsets and lengths must be 1-D and of equal length")
        return offs, lens

    def _ptr(self, t):
        return C.c_void_p(t.data_ptr())

    def upload(self, samples):
        """Pack host FASTQ byte strings into one device buffer at 16-byte aligned offsets;
        returns (tensor, offsets, lengths).  The bytes go through a pinned staging buffer that
        is kept (and grown) across calls, so the H2D copy is a single DMA at link speed."""
        torch = _torch()
        lens = np.array([len(s) for s in samples], dtype=np.uint64)
        offs = np.zeros(len(samples), dtype=np.uint64)
        pos = 0
        for i, n in enumerate(lens):
            offs[i] = pos
            pos += (int(n) + 15) // 16 * 16
        total = pos + 16
        pinned = getattr(self, "_pinned", None)
        if pinned is None or pinned.numel() < total:
            pinned = torch.empty(max(total, 1 << 20), dtype=torch.uint8, pin_memory=True)
            self._pinned = pinned
        host = pinned.numpy()
        for s, o, n in zip(samples, offs, lens):
            o, n = int(o), int(n)
            host[o:o + n] = np.frombuffer(s, dtype=np.uint8) if not isinstance(s, np.ndarray) else s
            host[o + n:(o + n + 15) // 16 * 16] = 0
        host[pos:total] = 0
        dev = torch.empty(total, dtype=torch.uint8, device=self.device)
        dev.copy_(pinned[:total], non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()   # the staging buffer is reused
        return dev, offs, lens

    def h2d_link_rate(self):
        """Bytes per second of a pinned -> device copy on this engine's device (128 MiB, best of three; measured once)."""
        if getattr(self, "_h2d_rate", None) is None:
            import time
            torch = _torch()
            n = 128 << 20
            src = torch.empty(n, dtype=torch.uint8).pin_memory()
            dst = torch.empty(n, dtype=torch.uint8, device=self.device)
            best = 0.0
            for _ in range(3):
                torch.cuda.synchronize(self.device)
                t0 = time.perf_counter()
                dst.copy_(src, non_blocking=True)
                torch.cuda.synchronize(self.device)
                best = max(best, n / (time.perf_counter() - t0))
            self._h2d_rate = best
            del src, dst
        return self._h2d_rate

    def stage_files(self, paths, pool=None, slot=0):
        """Host half of upload_files: read the files, AS THEY ARE ON DISK, into pinned staging buffer
        `slot` (kept and grown across calls) with parallel readinto, no intermediate copy.  Plain FASTQ
        files land at their final 16-byte aligned text offsets; gzip files (what step C of the
        reference writes, commands/image.py:696-708) stay compressed -- they cross PCIe that way and
        are inflated on the GPU by upload_staged -- and only get their text slot reserved (its size is
        the ISIZE word that ends a gzip file).  May run on a thread of its own (a pipeline stages the
        next batch while this one is copied and processed): the only GPU-runtime call is the pinned
        allocation, made with this engine's device current.  A file that cannot be read is staged with
        length 0 -- the caller sees an empty histogram for it and skips it as the reference skips a
        file dsk fails on (commands/image.py:1070-1075) -- and never takes the batch with it."""
        import os
        torch = _torch()
        torch.cuda.set_device(self.device)  # thread-local: a fresh thread would otherwise allocate on device 0

        def probe(p):
            """(is_gzip, bytes on disk, text bytes expected)"""
            try:
                size = os.path.getsize(p)
                with open(p, "rb") as f:
                    gz = f.read(2) == b"\x1f\x8b"
                    if not gz:
                        return False, size, size
                    if size < 18:
                        return True, 0, 0
                    f.seek(size - 4)
                    return True, size, int.from_bytes(f.read(4), "little")
            except OSError as e:
                print("cannot read", str(p) + ":", repr(e), file=sys.stderr)
                return False, 0, 0
        mapper = pool.map if pool is not None else map
        info = list(mapper(probe, paths))
        n = len(paths)
        is_gz = np.array([g for g, _, _ in info], dtype=bool)
        disk = np.array([d for _, d, _ in info], dtype=np.uint64)
        unreadable = np.zeros(n, dtype=bool)   # files that exist on the command line but could not be read
        for i, p in enumerate(paths):
            if int(disk[i]) == 0:
                try:
                    unreadable[i] = os.path.getsize(p) != 0    # (a gzip file under 18 bytes is not one)
                except OSError:
                    unreadable[i] = True
        lens = np.array([0 if g else t for g, _, t in info], dtype=np.uint64)      # gzip: known after the inflate
        caps = np.array([t for _, _, t in info], dtype=np.uint64)                  # text slot sizes
        for i in np.flatnonzero(is_gz):
            caps[i] = min(int(caps[i]), GZ_MAX_FIRST_RATIO * int(disk[i]) + (1 << 16))
        # staging layout: the plain files at their final 16-byte aligned text offsets, then the compressed files
        offs = np.zeros(n, dtype=np.uint64)
        pos = 0
        for i in np.flatnonzero(~is_gz):
            offs[i] = pos
            pos += (int(caps[i]) + 15) // 16 * 16
        plain_total = pos
        src = np.zeros(n, dtype=np.uint64)
        src[~is_gz] = offs[~is_gz]
        for i in np.flatnonzero(is_gz):
            src[i] = pos
            pos += (int(disk[i]) + 15) // 16 * 16
        # Plain files are not read at all where the platform allows it: they are mapped (MAP_SHARED, populated here,
        # on the staging threads) and upload_staged has the GPU copy them straight out of the page cache
        # (vk_upload_mapped) -- the read() into the pinned buffer, at ~3 GB/s per core, was what bounded plain-text
        # input, not the 57 GB/s link.  A file that cannot be mapped goes through the buffer as before.
        mapped = {}
        threads = getattr(pool, "_max_workers", 1) if pool is not None else 1
        if plain_route(threads, self) == "mapped":
            def map_file(i):
                if int(disk[i]) == 0:
                    return None
                if is_gz[i] and (int(disk[i]) < 28 or int(disk[i]) % 4096 == 0 or int(disk[i]) % 4096 > 4032):
                    return None   # (the inflate kernels read a compressed file in place: its last page keeps room behind the end)
                try:
                    fd = os.open(paths[i], os.O_RDONLY)
                    try:
                        if os.fstat(fd).st_size != int(disk[i]):
                            return None
                        mm = mmap.mmap(fd, int(disk[i]), flags=mmap.MAP_SHARED | getattr(mmap, "MAP_POPULATE", 0),
                                       prot=mmap.PROT_READ)
                    finally:
                        os.close(fd)
                except (OSError, ValueError):
                    return None
                view = np.frombuffer(mm, dtype=np.uint8)
                # pinned for the DMA engines here, ahead of the copy and beside it (upload_staged pins what is not)
                ctx = getattr(self, "ctx", None)
                pinned_now = bool(ctx) and self.L.vk_host_register(ctx, C.c_void_p(view.ctypes.data), view.size) == _capi.VK_OK
                if is_gz[i] and not pinned_now:   # kernels cannot read what is not pinned: through the staging buffer
                    view = None
                    mm.close()
                    return None
                return [mm, view, pinned_now]
            for i, m in enumerate(mapper(map_file, range(n))):
                if m is not None:
                    mapped[i] = m
        stage_total = pos + 16
        slots = self.__dict__.setdefault("_pinned_slots", {})
        pinned = slots.get(slot)
        if pinned is None or pinned.numel() < stage_total:
            pinned = torch.empty(max(stage_total, 1 << 20), dtype=torch.uint8, pin_memory=True)
            slots[slot] = pinned
        host = pinned.numpy()
        bgzf = {}    # file index -> member table of a BGZF file

        def fill(i):
            o, nb = int(src[i]), int(disk[i])
            got = 0
            if i in mapped:
                view = mapped[i][1]
                if is_gz[i] and view[3] & 4:
                    m = bgzf_members(view)
                    if m is not None:
                        caps[i] = int(m[2].sum())
                        bgzf[i] = m
                return
            if nb:
                try:
                    with open(paths[i], "rb") as f:
                        got = f.readinto(memoryview(host[o:o + nb]))
                except OSError as e:
                    print("cannot read", str(paths[i]) + ":", repr(e), file=sys.stderr)
                    got = -1
                if got != nb:      # unreadable or changed under us: an empty sample, not a dead batch
                    disk[i] = 0
                    lens[i] = 0
                    caps[i] = 0
                    unreadable[i] = True
                elif is_gz[i] and nb >= 28 and host[o + 3] & 4:
                    # many-member files (BGZF): the text size is the sum over the members, found by walking
                    # the block headers -- not the last member's size word (0 for BGZF's empty end marker)
                    m = bgzf_members(host[o:o + nb])
                    if m is not None:
                        caps[i] = int(m[2].sum())
                        bgzf[i] = m
            host[o + nb:(o + nb + 15) // 16 * 16] = 0
        list(mapper(fill, range(n)))
        # text layout in HBM: the plain region as staged, then the slots of the gzip files
        pos = plain_total
        for i in np.flatnonzero(is_gz):
            offs[i] = pos
            pos += (int(caps[i]) + 15) // 16 * 16
        text_total = pos + 16
        return {"pinned": pinned, "plain_total": plain_total, "stage_total": stage_total, "text_total": text_total,
                "is_gz": is_gz, "src": src, "disk": disk, "offs": offs, "lens": lens, "caps": caps,
                "paths": [str(p) for p in paths], "bgzf": bgzf, "mapped": mapped, "unreadable": unreadable,
                "plain_route": "mapped" if any(not is_gz[i] for i in mapped) else "staged"}

    def _release_mapped(self, mapped, which=None):
        """Unpin and unmap files of a batch (all, or those named) -- on a helper thread: it is a few milliseconds per
        batch that nothing has to wait for (close() waits for it)."""
        keys = list(mapped) if which is None else list(which)
        items = [mapped.pop(i) for i in keys]
        ctx, L = self.ctx, self.L

        def work():
            for m in items:
                mm, view, pinned_now = m
                if pinned_now:
                    L.vk_host_unregister(ctx, C.c_void_p(view.ctypes.data))
                m[1] = view = None
                try:
                    mm.close()
                except BufferError:
                    pass
        ex = self.__dict__.get("_releaser")
        if ex is None:
            from concurrent.futures import ThreadPoolExecutor
            ex = self.__dict__["_releaser"] = ThreadPoolExecutor(1)
        ex.submit(work)

    def upload_staged(self, staged, timings=None):
        """Device half: one H2D DMA of the plain text, one of the compressed files, and the gzip files
        inflated in HBM into their text slots (vk_inflate_device).  Returns (tensor, offsets, lengths);
        the staging buffer may be refilled once this returns.  A gzip file the GPU rejects (bad header
        or data, truncated, size word or CRC-32 wrong) gets length 0 and a line on stderr."""
        mapped = staged.get("mapped") or {}
        try:
            return self._upload_staged(staged, mapped, timings)
        finally:
            if mapped:      # also when a call in there raised: the files' pages must not stay pinned
                self._release_mapped(mapped)

    def _upload_staged(self, staged, mapped, timings):
        torch = _torch()
        pinned, plain_total, stage_total = staged["pinned"], staged["plain_total"], staged["stage_total"]
        offs, lens, is_gz = staged["offs"], staged["lens"].copy(), staged["is_gz"]
        # per file: 0 = its text is in HBM (possibly empty), VK_GZ_* bits of a failed inflate, 0x100 = unreadable
        fstat = np.where(staged.get("unreadable", np.zeros(len(offs), dtype=bool)), 0x100, 0).astype(np.uint32)
        self.last_upload_status = fstat
        idx = [i for i in sorted(mapped) if not is_gz[i]]
        if idx:
            # (zeroed: a mapped file brings no padding up to its 16-byte rounded end along)
            dev = torch.zeros(staged["text_total"], dtype=torch.uint8, device=self.device)
            views = [mapped[i][1] for i in idx]
            srcp = (C.c_void_p * len(idx))(*[v.ctypes.data for v in views])
            flags = np.array([1 if mapped[i][2] else 0 for i in idx], dtype=np.uint8)
            st = np.zeros(len(idx), dtype=np.uint32)
            _capi.check(self.ctx, self.L.vk_upload_mapped(
                self.ctx, self._ptr(dev), np.ascontiguousarray(offs[idx]).ctypes.data_as(C.POINTER(C.c_uint64)), srcp,
                np.ascontiguousarray(staged["disk"][idx]).ctypes.data_as(C.POINTER(C.c_uint64)),
                flags.ctypes.data_as(C.POINTER(C.c_uint8)), len(idx), st.ctypes.data_as(C.POINTER(C.c_uint32))),
                "vk_upload_mapped")
            for j, i in enumerate(idx):
                if st[j]:   # the pages could not be registered: through a pageable copy, once
                    dev[int(offs[i]):int(offs[i]) + views[j].size].copy_(torch.from_numpy(views[j].copy()))
            del views, srcp
            self._release_mapped(mapped, idx)
            for i in np.flatnonzero(~is_gz):           # the files that could not be mapped lie in the staging buffer
                if int(i) not in idx and int(staged["disk"][i]):
                    o, nb = int(offs[i]), int(staged["disk"][i])
                    dev[o:o + nb].copy_(pinned[o:o + nb], non_blocking=True)
        else:
            dev = torch.empty(staged["text_total"], dtype=torch.uint8, device=self.device)
            if plain_total:
                dev[:plain_total].copy_(pinned[:plain_total], non_blocking=True)
        gi = np.flatnonzero(is_gz & (staged["disk"] > 0))
        if gi.size:
            # the compressed bytes are not copied: the inflate kernels read them where they are, in the pinned
            # staging buffer, over PCIe (each byte once by the block-start finder and once by the decoder,
            # ~12 GB/s of a 57 GB/s link) -- 27 ms of H2D copy per 1.5 GB that nothing had to wait for
            # ... or, for a rank with few I/O threads, in the files' own page-cache pages (mapped and pinned by stage_files):
            # every file's address, as an offset from the lowest one
            addr = np.array([mapped[int(i)][1].ctypes.data if int(i) in mapped else pinned.data_ptr() + int(staged["src"][i])
                             for i in gi], dtype=np.uint64)
            gzdev = int(addr.min())
            rel = addr - np.uint64(gzdev)
            import time
            ti = time.perf_counter()
            # A BGZF file goes in as its members: no member refers to another, so each is a gzip file of its
            # own -- a wavefront per member instead of one per file, its size and CRC-32 checked like any file's.
            table = staged.get("bgzf") or {}
            go, gl, oo, oc, owner = [], [], [], [], []
            for j, i in enumerate(gi):
                base = rel[j]
                if int(i) in table:
                    mo, ms, mt = table[int(i)]
                    ends = np.cumsum(mt)
                    go.append(base + mo)
                    gl.append(ms)
                    oo.append(offs[i] + ends - mt)
                    oc.append(mt)
                    owner.append(np.full(mo.size, j, dtype=np.int64))
                else:
                    go.append(np.array([base], dtype=np.uint64))
                    gl.append(staged["disk"][i:i + 1])
                    oo.append(offs[i:i + 1])
                    oc.append(staged["caps"][i:i + 1])
                    owner.append(np.array([j], dtype=np.int64))
            owner = np.concatenate(owner)
            g1, s1 = self.inflate(gzdev, np.concatenate(go), np.concatenate(gl), dev, np.concatenate(oo), np.concatenate(oc))
            got = np.zeros(gi.size, dtype=np.uint64)
            st = np.zeros(gi.size, dtype=np.uint32)
            np.add.at(got, owner, g1)
            np.bitwise_or.at(st, owner, s1)
            if timings is not None:
                timings["inflate_s"] = timings.get("inflate_s", 0.0) + time.perf_counter() - ti
            # More text than the slot held (several members and only the last one's size word known, or a
            # file that expands past GZ_MAX_FIRST_RATIO): inflate those again into a side buffer -- of the
            # size the first pass reported where it could (the chunked path decodes the whole file before it
            # looks at the slot), else of 32x, 256x and finally DEFLATE's limit of 1032x the compressed size.
            tried = {j: int(staged["caps"][gi[j]]) for j in range(gi.size)}   # the slot size of each file's last attempt
            for grow in (32, 256, 1032):
                over = [j for j in range(gi.size) if st[j] == _capi.VK_GZ_OVERFLOW]
                if not over:
                    break
                oi = gi[over]
                # (a length beyond the slot is the size the call reported as needed; one within it is what fitted)
                # -- never a slot that is not larger than the one that just overflowed (a small file's first slot,
                # 64x its size, is above the ladder's first steps)
                c2 = np.array([int(got[jj]) if int(got[jj]) > tried[jj] else
                               max(int(staged["disk"][gi[jj]]) * grow + (1 << 16), 2 * tried[jj]) for jj in over], dtype=np.uint64)
                for j, jj in enumerate(over):
                    tried[jj] = int(c2[j])
                o2 = np.zeros(oi.size, dtype=np.uint64)
                p = 0
                for j in range(oi.size):
                    o2[j] = p
                    p += (int(c2[j]) + 15) // 16 * 16
                try:
                    side = torch.empty(p + 16, dtype=torch.uint8, device=self.device)
                except RuntimeError as e:       # no room for that much text: these files fail, the batch lives
                    print("gzip inflate: no memory for", p, "bytes of text:", repr(e)[:120], file=sys.stderr)
                    break
                g2, s2 = self.inflate(gzdev, rel[over], staged["disk"][oi], side, o2, c2)
                base = dev.numel()
                dev = torch.cat([dev, side])
                del side
                for j, jj in enumerate(over):
                    got[jj], st[jj] = g2[j], s2[j]
                    offs = offs.copy() if offs is staged["offs"] else offs
                    offs[oi[j]] = base + int(o2[j])
            for j, i in enumerate(gi):
                if st[j]:
                    print(f"gzip inflate failed (status {int(st[j])}):", staged["paths"][i], file=sys.stderr)
                    fstat[i] = st[j]
                else:
                    lens[i] = got[j]
            if mapped:
                self._release_mapped(mapped)   # (every inflate call has synchronised)
        torch.cuda.current_stream(self.device).synchronize()
        if mapped:
            self._release_mapped(mapped)
        return dev, offs, lens

    def upload_files(self, paths, pool=None):
        """Like upload() for files on disk (stage_files + upload_staged)."""
        return self.upload_staged(self.stage_files(paths, pool))

    def inflate(self, gz, gz_offsets, gz_lengths, out, out_offsets, out_caps):
        """gzip files -> FASTQ text in HBM (vk_inflate_device; dsk reads .gz natively,
        commands/image.py:771-790).  gz: uint8 tensor in device memory or in pinned host memory (the
        kernels read it in place); out: uint8 device tensor; offsets / lengths / caps: uint64 arrays.  Returns (text_lengths uint64[n], status uint32[n] of VK_GZ_* bits); synchronises."""
        n = len(gz_offsets)
        go, gl = self._desc(gz_offsets, gz_lengths)
        oo, oc = self._desc(out_offsets, out_caps)
        lens = np.zeros(n, dtype=np.uint64)
        status = np.zeros(n, dtype=np.uint32)
        gzp = C.c_void_p(gz) if isinstance(gz, int) else self._ptr(gz)   # (an address: host memory pinned for the GPU)
        st = self.L.vk_inflate_device(self.ctx, gzp, go.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      gl.ctypes.data_as(C.POINTER(C.c_uint64)), n, self._ptr(out),
                                      oo.ctypes.data_as(C.POINTER(C.c_uint64)), oc.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      lens.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      status.ctypes.data_as(C.POINTER(C.c_uint32)))
        _capi.check(self.ctx, st, "vk_inflate_device")
        return lens, status

    # -- stages ----------------------------------------------------------------
    def count(self, fastq, offsets, lengths, parts=0, hist=None, status=None):
        """K1 (+check): forward-strand histograms int32/uint32 [n, 4^k] and status [n]."""
        torch = _torch()
        offs, lens = self._desc(offsets, lengths)
        n = len(offs)
        if hist is None:
            hist = torch.empty((n, self.ncode), dtype=torch.int32, device=self.device)
        if status is None:
            status = torch.empty((n,), dtype=torch.int32, device=self.device)
        st = self.L.vk_count_device(self.ctx, self._ptr(fastq), offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                                    lens.ctypes.data_as(C.POINTER(C.c_uint64)), n, self.k, parts,
                                    self._ptr(hist), self._ptr(status))
        _capi.check(self.ctx, st, "vk_count_device")
        return hist, status

    def count_sampled(self, fastq, offsets, lengths, seeds, thresholds, parts=0, hist=None, status=None,
                      sites=None):
        """K1 over a pseudo-random subset of each sample's reads (vk_count_sampled_device):
        seeds / thresholds are per-sample host arrays (threshold = fraction * 2^32, 2^32 = all).
        Returns (hist [n, 4^k], status [n], sites int64 [n, 2] = (bytes of all sequence lines,
        bytes of the sequence lines of the reads taken))."""
        torch = _torch()
        offs, lens = self._desc(offsets, lengths)
        n = len(offs)
        seeds = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint64), (n,)))
        thr = np.ascontiguousarray(np.broadcast_to(np.asarray(thresholds, dtype=np.uint64), (n,)))
        if hist is None:
            hist = torch.empty((n, self.ncode), dtype=torch.int32, device=self.device)
        if status is None:
            status = torch.empty((n,), dtype=torch.int32, device=self.device)
        if sites is None:
            sites = torch.empty((n, 2), dtype=torch.int64, device=self.device)
        u64p = C.POINTER(C.c_uint64)
        st = self.L.vk_count_sampled_device(self.ctx, self._ptr(fastq), offs.ctypes.data_as(u64p),
                                            lens.ctypes.data_as(u64p), n, self.k, parts, seeds.ctypes.data_as(u64p),
                                            thr.ctypes.data_as(u64p), self._ptr(hist), self._ptr(status),
                                            self._ptr(sites))
        _capi.check(self.ctx, st, "vk_count_sampled_device")
        return hist, status, sites

    def read_index(self, fastq, offsets, lengths, parts=0):
        """The read index of a batch (vk_read_index_device): (nsites uint64[n], status uint32[n]) on the host; the
        context keeps the index, and count_sampled calls on these samples walk the reads they take."""
        offs, lens = self._desc(offsets, lengths)
        n = len(offs)
        sites = np.zeros(n, dtype=np.uint64)
        status = np.zeros(n, dtype=np.uint32)
        u64p = C.POINTER(C.c_uint64)
        st = self.L.vk_read_index_device(self.ctx, self._ptr(fastq), offs.ctypes.data_as(u64p), lens.ctypes.data_as(u64p), n, parts,
                                         sites.ctypes.data_as(u64p), status.ctypes.data_as(C.POINTER(C.c_uint32)))
        _capi.check(self.ctx, st, "vk_read_index_device")
        return sites, status

    def count_index(self, fastq, offsets, lengths, parts=0, hist=None, status=None):
        """count() and read_index() in one pass over the text (vk_count_index_device):
        (hist [n, 4^k] on the device, nsites uint64[n], status uint32[n] on the host)."""
        torch = _torch()
        offs, lens = self._desc(offsets, lengths)
        n = len(offs)
        if hist is None:
            hist = torch.empty((n, self.ncode), dtype=torch.int32, device=self.device)
        if status is None:
            status = torch.empty((n,), dtype=torch.int32, device=self.device)
        sites = np.zeros(n, dtype=np.uint64)
        st_h = np.zeros(n, dtype=np.uint32)
        u64p = C.POINTER(C.c_uint64)
        st = self.L.vk_count_index_device(self.ctx, self._ptr(fastq), offs.ctypes.data_as(u64p), lens.ctypes.data_as(u64p), n, self.k,
                                          parts, self._ptr(hist), self._ptr(status), sites.ctypes.data_as(u64p),
                                          st_h.ctypes.data_as(C.POINTER(C.c_uint32)))
        _capi.check(self.ctx, st, "vk_count_index_device")
        return hist, sites, st_h

    def images(self, hist, img=None):
        """K2: uint8 images [n, side, side] from histograms [n, 4^k]."""
        torch = _torch()
        n = hist.shape[0]
        if img is None:
            img = torch.empty((n, self.side, self.side), dtype=torch.uint8, device=self.device)
        st = self.L.vk_image_device(self.ctx, self._ptr(hist), n, self.k, self._ptr(img))
        _capi.check(self.ctx, st, "vk_image_device")
        return img

    def fastq_to_images(self, fastq, offsets, lengths, parts=0, hist=None, status=None, img=None):
        torch = _torch()
        offs, lens = self._desc(offsets, lengths)
        n = len(offs)
        if hist is None:
            hist = torch.empty((n, self.ncode), dtype=torch.int32, device=self.device)
        if status is None:
            status = torch.empty((n,), dtype=torch.int32, device=self.device)
        if img is None:
            img = torch.empty((n, self.side, self.side), dtype=torch.uint8, device=self.device)
        st = self.L.vk_fastq_to_image_device(self.ctx, self._ptr(fastq),
                                             offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                                             lens.ctypes.data_as(C.POINTER(C.c_uint64)), n, self.k, parts,
                                             self._ptr(hist), self._ptr(status), self._ptr(img))
        _capi.check(self.ctx, st, "vk_fastq_to_image_device")
        return img, hist, status

    def synth(self, sample0, nsamples, reads, readlen=150, seed=20250824, dist=0, out=None):
        """Synthetic FASTQ for samples sample0..sample0+nsamples-1, generated in HBM (dist 0, 1: records of one
        size; dist 2: reads shaped like fastp's output, samples of different sizes -- synth.py)."""
        torch = _torch()
        u64p = C.POINTER(C.c_uint64)
        if dist == 2:
            lens = np.zeros(nsamples, dtype=np.uint64)
            _capi.check(self.ctx, self.L.vk_synth_shaped_lengths(self.ctx, sample0, nsamples, reads, readlen, C.c_uint64(seed),
                                                                  lens.ctypes.data_as(u64p)), "vk_synth_shaped_lengths")
            offs = np.zeros(nsamples, dtype=np.uint64)
            if nsamples > 1:
                offs[1:] = np.cumsum((lens[:-1] + np.uint64(15)) // np.uint64(16) * np.uint64(16))
            total = int(offs[-1] + lens[-1]) if nsamples else 0
            if out is None or out.numel() < (total + 15) // 16 * 16 + 16:
                out = torch.empty(((total + 15) // 16 * 16 + 16,), dtype=torch.uint8, device=self.device)
            _capi.check(self.ctx, self.L.vk_synth_shaped_device(self.ctx, self._ptr(out), offs.ctypes.data_as(u64p), sample0,
                                                                 nsamples, reads, readlen, C.c_uint64(seed)), "vk_synth_shaped_device")
            return out, offs, lens
        rec = 2 * readlen + 20
        total = rec * reads * nsamples
        if out is None:
            out = torch.empty(((total + 15) // 16 * 16 + 16,), dtype=torch.uint8, device=self.device)
        st = self.L.vk_synth_fastq_device(self.ctx, self._ptr(out), sample0, nsamples, reads, readlen,
                                          C.c_uint64(seed), dist)
        _capi.check(self.ctx, st, "vk_synth_fastq_device")
        offs = np.arange(nsamples, dtype=np.uint64) * np.uint64(rec * reads)
        lens = np.full(nsamples, rec * reads, dtype=np.uint64)
        return out, offs, lens

    def last_count_launch(self):
        g, b, l = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self.L.vk_last_count_launch(self.ctx, C.byref(g), C.byref(b), C.byref(l))
        return {"grid": g.value, "block": b.value, "lds_bytes": l.value}

    def last_count_general(self):
        """(pieces of the last k <= 7 count that took the general path, pieces in all)"""
        g, n = C.c_uint64(), C.c_uint64()
        _capi.check(self.ctx, self.L.vk_last_count_general(self.ctx, C.byref(g), C.byref(n)), "vk_last_count_general")
        return g.value, n.value

    # -- host conveniences -------------------------------------------------------
    def count_host(self, data):
        """One host FASTQ byte string -> (hist uint32[4^k], status bits)."""
        buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
        hist = np.empty(self.ncode, dtype=np.uint32)
        stw = C.c_uint32(0)
        st = self.L.vk_count_host(self.ctx, C.c_void_p(buf.ctypes.data if buf.size else 0), buf.size, self.k,
                                  hist.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(stw))
        if st not in (_capi.VK_OK, _capi.VK_EFORMAT):
            _capi.check(self.ctx, st, "vk_count_host")
        return hist, stw.value

    def image_host(self, hist):
        hist = np.ascontiguousarray(hist, dtype=np.uint32)
        img = np.empty(self.npix, dtype=np.uint8)
        st = self.L.vk_image_host(self.ctx, hist.ctypes.data_as(C.POINTER(C.c_uint32)), self.k,
                                  img.ctypes.data_as(C.POINTER(C.c_uint8)))
        _capi.check(self.ctx, st, "vk_image_host")
        return img.reshape(self.side, self.side)


def count_giant_sample(engine, data, rank=0, world=1):
    """ONE sample spread over the ranks of a torch.distributed job (optional row): every
    rank counts a record-aligned byte range of the FASTQ text on its own GPU, then the 4^k u32
    histograms are summed with one all-reduce (RCCL when the group is nccl).  Returns the full
    histogram tensor (identical on every rank) and this rank's status word."""
    from .shard import allreduce_sum_, split_at_records, widen_u32
    start, end = split_at_records(data, world)[rank]
    part = data[start:end]
    dev, offs, lens = engine.upload([part])
    hist, status = engine.count(dev, offs, lens)
    h = widen_u32(hist.view(-1))              # sum in 64 bit (unsigned widening): counts of a giant sample may pass 2^31
    allreduce_sum_(h)
    return h, int(status.cpu()[0])
