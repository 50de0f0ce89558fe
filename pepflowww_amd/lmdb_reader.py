"""Read-only access to an LMDB environment file in pure Python (mmap + struct), for the dataset cache the reference keeps
(`models_con/pep_dataloader.py:87-196`: `lmdb.open(path, subdir=False, readonly=True, lock=False)`, keys = entry ids, values =
pickled dicts).  The `lmdb` module (and liblmdb) are absent from this environment, so this is written from the published on-disk
format of LMDB 0.9 (lmdb.h / mdb.c: MDB_meta, MDB_db, MDB_page, MDB_node) and is verified here only against databases
assembled by `tests/lmdb_fixture.py` from the same description -- FORMAT PARITY WITH liblmdb IS UNPINNED until it is run
against a file written by the real library.

Supported: one environment file, the main (unnamed) database, default key order (memcmp), values on overflow pages
(F_BIGDATA); native little-endian, 64-bit page numbers.  Not supported (raises): named sub-databases, MDB_DUPSORT values.

Layout used (all integers little-endian):
  page header (16 B)  : pgno u64 | pad u16 | flags u16 | lower u16, upper u16  (overflow pages: u32 page count instead)
  page flags          : P_BRANCH 0x01, P_LEAF 0x02, P_OVERFLOW 0x04, P_META 0x08, P_LEAF2 0x20
  meta page (0 and 1) : header | magic u32 0xBEEFC0DE | version u32 | address u64 | mapsize u64 | MDB_db free | MDB_db main |
                        last_pg u64 | txnid u64;  the meta with the larger txnid is current; page size = free.md_pad
  MDB_db (48 B)       : pad u32 | flags u16 | depth u16 | branch_pages u64 | leaf_pages u64 | overflow_pages u64 | entries u64 | root u64
  node (8 B + key..)  : lo u16 | hi u16 | flags u16 | ksize u16 | key | data
                        leaf: data size = lo | hi << 16; F_BIGDATA 0x01: data = u64 page number of an overflow run
                        branch: child page = lo | hi << 16 | flags << 32 (node 0 has an empty key)
  node pointers       : u16 offsets from the page start, (lower - 16) / 2 of them, right after the header
"""
import mmap
import struct

MAGIC, P_BRANCH, P_LEAF, P_OVERFLOW, P_META, P_LEAF2 = 0xBEEFC0DE, 0x01, 0x02, 0x04, 0x08, 0x20
F_BIGDATA, F_SUBDATA, F_DUPDATA = 0x01, 0x02, 0x04
PAGEHDRSZ, P_INVALID = 16, 0xFFFFFFFFFFFFFFFF
_META = struct.Struct("<IIQQ")            # magic, version, address, mapsize
_DB = struct.Struct("<IHHQQQQQ")          # pad, flags, depth, branch, leaf, overflow, entries, root
_NODE = struct.Struct("<HHHH")            # lo, hi, flags, ksize


class LmdbFormatError(ValueError):
    pass


class LmdbReader:
    """`with LmdbReader(path) as db: db.keys(); db.get(key)` -- keys and values are bytes."""

    def __init__(self, path):
        self._f = open(path, "rb")
        self._m = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        metas = []
        psize = None
        for pg in (0, 1):
            off = pg * (psize or 4096)
            if off + PAGEHDRSZ + _META.size + 2 * _DB.size + 16 > len(self._m):
                break
            flags = struct.unpack_from("<H", self._m, off + 10)[0]
            magic, version, _, _ = _META.unpack_from(self._m, off + PAGEHDRSZ)
            if magic != MAGIC or not flags & P_META:
                if pg == 0:
                    raise LmdbFormatError("not an LMDB environment (bad magic / meta flag on page 0)")
                continue
            if version != 1:
                raise LmdbFormatError(f"unsupported LMDB data version {version}")
            free = _DB.unpack_from(self._m, off + PAGEHDRSZ + _META.size)
            main = _DB.unpack_from(self._m, off + PAGEHDRSZ + _META.size + _DB.size)
            last_pg, txnid = struct.unpack_from("<QQ", self._m, off + PAGEHDRSZ + _META.size + 2 * _DB.size)
            psize = psize or free[0]
            metas.append((txnid, main, last_pg))
        if not metas:
            raise LmdbFormatError("no valid meta page")
        self.page_size = psize
        _, main, self.last_pg = max(metas, key=lambda t: t[0])
        self.entries, self._root, self._dbflags = main[6], main[7], main[1]
        if self._dbflags & 0x04:                       # MDB_DUPSORT
            raise LmdbFormatError("MDB_DUPSORT databases are not supported")

    # ---- pages ----
    def _page(self, pgno):
        off = pgno * self.page_size
        if off + PAGEHDRSZ > len(self._m):
            raise LmdbFormatError(f"page {pgno} beyond the end of the file")
        flags, lower, upper = struct.unpack_from("<HHH", self._m, off + 10)
        return off, flags, lower

    def _nodes(self, off, lower):
        n = (lower - PAGEHDRSZ) // 2
        return struct.unpack_from(f"<{n}H", self._m, off + PAGEHDRSZ) if n else ()

    def _leaf_value(self, off, ptr):
        lo, hi, nflags, ksize = _NODE.unpack_from(self._m, off + ptr)
        key = bytes(self._m[off + ptr + 8: off + ptr + 8 + ksize])
        dsize = lo | (hi << 16)
        if nflags & (F_SUBDATA | F_DUPDATA):
            raise LmdbFormatError("sub-database / duplicate nodes are not supported")
        dpos = off + ptr + 8 + ksize
        if nflags & F_BIGDATA:
            (ovpg,) = struct.unpack_from("<Q", self._m, dpos)
            ooff, oflags, _ = self._page(ovpg)
            if not oflags & P_OVERFLOW:
                raise LmdbFormatError(f"page {ovpg} is not an overflow page")
            dpos = ooff + PAGEHDRSZ
        return key, (dpos, dsize)

    def _walk(self, pgno):
        off, flags, lower = self._page(pgno)
        if flags & P_LEAF:
            if flags & P_LEAF2:
                raise LmdbFormatError("LEAF2 pages (MDB_DUPFIXED) are not supported")
            for ptr in self._nodes(off, lower):
                yield self._leaf_value(off, ptr)
        elif flags & P_BRANCH:
            for ptr in self._nodes(off, lower):
                lo, hi, nflags, _ = _NODE.unpack_from(self._m, off + ptr)
                yield from self._walk(lo | (hi << 16) | (nflags << 32))
        else:
            raise LmdbFormatError(f"page {pgno}: unexpected flags {flags:#x}")

    # ---- API ----
    def items(self):
        """(key, value) pairs in key order."""
        if self._root == P_INVALID:
            return
        for key, (pos, size) in self._walk(self._root):
            yield key, bytes(self._m[pos: pos + size])

    def keys(self):
        if self._root == P_INVALID:
            return []
        return [k for k, _ in self._walk(self._root)]

    def get(self, key, default=None):
        """B+tree descent (memcmp order), like mdb_get."""
        if self._root == P_INVALID:
            return default
        pgno = self._root
        while True:
            off, flags, lower = self._page(pgno)
            ptrs = self._nodes(off, lower)
            if flags & P_LEAF:
                lo_i, hi_i = 0, len(ptrs)
                while lo_i < hi_i:                     # binary search over the sorted leaf
                    mid = (lo_i + hi_i) // 2
                    ksize = struct.unpack_from("<H", self._m, off + ptrs[mid] + 6)[0]
                    k = bytes(self._m[off + ptrs[mid] + 8: off + ptrs[mid] + 8 + ksize])
                    if k < key:
                        lo_i = mid + 1
                    else:
                        hi_i = mid
                if lo_i < len(ptrs):
                    k, (pos, size) = self._leaf_value(off, ptrs[lo_i])
                    if k == key:
                        return bytes(self._m[pos: pos + size])
                return default
            if not flags & P_BRANCH:
                raise LmdbFormatError(f"page {pgno}: unexpected flags {flags:#x}")
            child = 0                                  # last node whose key <= search key (node 0: empty key = -inf)
            for i in range(1, len(ptrs)):
                ksize = struct.unpack_from("<H", self._m, off + ptrs[i] + 6)[0]
                k = bytes(self._m[off + ptrs[i] + 8: off + ptrs[i] + 8 + ksize])
                if k <= key:
                    child = i
                else:
                    break
            lo, hi, nflags, _ = _NODE.unpack_from(self._m, off + ptrs[child])
            pgno = lo | (hi << 16) | (nflags << 32)

    def __len__(self):
        return self.entries

    def close(self):
        if self._m is not None:
            self._m.close()
            self._f.close()
            self._m = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
