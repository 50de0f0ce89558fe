"""Test-only writer of a minimal LMDB environment file (one main database, default key order), assembled from the on-disk format
description in pepflowww_amd/lmdb_reader.py: leaf pages, branch pages (any depth), overflow runs for large values, two meta pages.
liblmdb is not available here, so this is NOT a check against the real library -- it exercises every page / node kind the
reader handles."""
import struct

PAGEHDRSZ, P_BRANCH, P_LEAF, P_OVERFLOW, P_META, F_BIGDATA = 16, 0x01, 0x02, 0x04, 0x08, 0x01
P_INVALID = 0xFFFFFFFFFFFFFFFF


def _even(n):
    return (n + 1) & ~1


def _page(psize, pgno, flags, nodes):
    """nodes: list of node byte strings (header + key + data), placed from the page end downwards, pointers ascending."""
    buf = bytearray(psize)
    upper = psize
    ptrs = []
    for nd in nodes:
        upper -= _even(len(nd))
        buf[upper: upper + len(nd)] = nd
        ptrs.append(upper)
    lower = PAGEHDRSZ + 2 * len(ptrs)
    assert lower <= upper, "page overflow in the fixture writer"
    struct.pack_into("<QHHHH", buf, 0, pgno, 0, flags, lower, upper)
    struct.pack_into(f"<{len(ptrs)}H", buf, PAGEHDRSZ, *ptrs)
    return bytes(buf)


def write_lmdb(path, items, psize=4096):
    items = sorted(items.items())
    pages = {}                                   # pgno -> bytes
    next_pg = 2
    nodemax = (psize - PAGEHDRSZ) // 2 - 2       # like mdb's me_nodemax: at least two nodes per page
    n_over = 0
    leaf_nodes = []
    for k, v in items:
        if 8 + len(k) + len(v) > nodemax:        # value on an overflow run
            npg = (PAGEHDRSZ + len(v) + psize - 1) // psize
            run = bytearray(npg * psize)
            struct.pack_into("<QHHI", run, 0, next_pg, 0, P_OVERFLOW, npg)
            run[PAGEHDRSZ: PAGEHDRSZ + len(v)] = v
            for i in range(npg):
                pages[next_pg + i] = bytes(run[i * psize: (i + 1) * psize])
            data = struct.pack("<Q", next_pg)
            next_pg += npg
            n_over += npg
            nflags = F_BIGDATA
        else:
            data, nflags = v, 0
        leaf_nodes.append((k, struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, nflags, len(k)) + k + data))

    def pack_level(nodes, flags):
        """Greedy fill; returns [(first_key, pgno)]."""
        nonlocal next_pg
        out, cur, used = [], [], PAGEHDRSZ
        for k, nd in nodes:
            need = _even(len(nd)) + 2
            if cur and used + need > psize:
                pages[next_pg] = _page(psize, next_pg, flags, [n for _, n in cur])
                out.append((cur[0][0], next_pg))
                next_pg += 1
                cur, used = [], PAGEHDRSZ
            cur.append((k, nd))
            used += need
        if cur:
            pages[next_pg] = _page(psize, next_pg, flags, [n for _, n in cur])
            out.append((cur[0][0], next_pg))
            next_pg += 1
        return out

    n_leaf = n_branch = 0
    depth = 0
    root = P_INVALID
    if leaf_nodes:
        level = pack_level(leaf_nodes, P_LEAF)
        n_leaf, depth = len(level), 1
        while len(level) > 1:
            bn = []
            for i, (k, pg) in enumerate(level):
                key = b"" if i == 0 else k
                bn.append((k, struct.pack("<HHHH", pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF, len(key)) + key))
            # node 0 of EVERY branch page has an empty key: re-pack page by page
            new_level, cur, used = [], [], PAGEHDRSZ

            def flush():
                nonlocal next_pg, cur, used
                k0 = cur[0][0]
                pg0 = struct.unpack("<HHHH", cur[0][1][:8])
                first = struct.pack("<HHHH", pg0[0], pg0[1], pg0[2], 0)
                pages[next_pg] = _page(psize, next_pg, P_BRANCH, [first] + [n for _, n in cur[1:]])
                new_level.append((k0, next_pg))
                next_pg += 1
                cur, used = [], PAGEHDRSZ
            for k, nd in bn:
                need = _even(len(nd)) + 2
                if cur and used + need > psize:
                    flush()
                if not cur:                        # this node becomes node 0: re-encode with its real key kept for the parent
                    pg = struct.unpack("<HHHH", nd[:8])
                    nd = struct.pack("<HHHH", pg[0], pg[1], pg[2], len(k)) + k
                cur.append((k, nd))
                used += need
            flush()
            n_branch += len(new_level)
            level = new_level
            depth += 1
        root = level[0][1]

    def meta(pgno, txnid, with_data):
        buf = bytearray(psize)
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, P_META, 0, 0)
        struct.pack_into("<IIQQ", buf, PAGEHDRSZ, 0xBEEFC0DE, 1, 0, 1 << 30)
        struct.pack_into("<IHHQQQQQ", buf, PAGEHDRSZ + 24, psize, 0, 0, 0, 0, 0, 0, P_INVALID)            # free DB
        if with_data:
            struct.pack_into("<IHHQQQQQ", buf, PAGEHDRSZ + 24 + 48, 0, 0, depth, n_branch, n_leaf, n_over, len(items), root)
            last = next_pg - 1
        else:
            struct.pack_into("<IHHQQQQQ", buf, PAGEHDRSZ + 24 + 48, 0, 0, 0, 0, 0, 0, 0, P_INVALID)
            last = 1
        struct.pack_into("<QQ", buf, PAGEHDRSZ + 24 + 96, last, txnid)
        return bytes(buf)

    pages[0] = meta(0, 0, False)
    pages[1] = meta(1, 1, True)
    with open(path, "wb") as f:
        for pg in range(next_pg):
            f.write(pages[pg])
