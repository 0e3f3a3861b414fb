"""Minimal self-contained HDF5 reader/writer (no h5py needed on the GPU box).

Covers exactly what Keras-2.0.9 weight files use (train.py:316-355 ModelCheckpoint,
model.py:119 load_weights): version-0 superblock, version-1 object headers, old-style
groups (symbol table: v1 B-tree + local heap + SNOD nodes), contiguous little-endian
numeric datasets plus chunked gzip/shuffle datasets (the batch blobs of
data/avc/sample.py:565-568), and attributes holding fixed-length byte strings (scalar or 1-D),
variable-length strings (read only; what h5py >= 3 writes) or numeric arrays.  Files written here open with libhdf5/h5py; files written by
h5py's default settings (libver earliest) read back here.  Anything else (chunked or
filtered datasets, v2 object headers, new-style link messages) raises H5Error.
"""
import os
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b'\x89HDF\r\n\x1a\n'
LEAF_K = 4          # symbol-table node holds 2*LEAF_K entries
INTERNAL_K = 16     # group B-tree node holds 2*INTERNAL_K children
FREE_NULL = 1       # H5HL_FREE_NULL: "no free block" marker libhdf5 writes/accepts

MSG_DATASPACE, MSG_DATATYPE, MSG_FILL, MSG_LAYOUT, MSG_ATTRIBUTE, MSG_CONT, MSG_STAB = 0x1, 0x3, 0x5, 0x8, 0xC, 0x10, 0x11


class H5Error(IOError):
    pass


class Group(object):
    def __init__(self):
        self.children = {}      # name -> Group | np.ndarray
        self.attrs = {}         # name -> bytes | np.ndarray

    def create_group(self, name):
        g = self
        for part in name.strip('/').split('/'):
            nxt = g.children.get(part)
            if nxt is None:
                nxt = Group()
                g.children[part] = nxt
            g = nxt
        return g

    def create_dataset(self, name, data, compression=None):
        parts = name.strip('/').split('/')
        g = self.create_group('/'.join(parts[:-1])) if len(parts) > 1 else self
        arr = np.ascontiguousarray(data)
        g.children[parts[-1]] = GzipArray(arr) if compression == 'gzip' else arr

    def __getitem__(self, name):
        g = self
        for part in name.strip('/').split('/'):
            g = g.children[part]
        return g


class GzipArray(object):
    """Marks a dataset to be written chunked + deflate (like `compression='gzip'` in h5py)."""

    def __init__(self, arr):
        self.arr = arr


def _pad8(b):
    return b + b'\0' * ((-len(b)) % 8)


# ------------------------------------------------------------------------------------------
# writer
# ------------------------------------------------------------------------------------------
class _Out(object):
    def __init__(self):
        self.buf = bytearray(96)          # superblock patched in at the end

    def alloc(self, data):
        while len(self.buf) % 8:
            self.buf.append(0)
        addr = len(self.buf)
        self.buf += data
        return addr


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == 'f' and dt.itemsize in (4, 8):
        if dt.itemsize == 4:
            props = struct.pack('<HHBBBBI', 0, 32, 23, 8, 0, 23, 127)
            bits = bytes([0x20, 31, 0])
        else:
            props = struct.pack('<HHBBBBI', 0, 64, 52, 11, 0, 52, 1023)
            bits = bytes([0x20, 63, 0])
        return bytes([0x11]) + bits + struct.pack('<I', dt.itemsize) + props
    if dt.kind in 'iu':
        bits = bytes([0x08 if dt.kind == 'i' else 0x00, 0, 0])
        return bytes([0x10]) + bits + struct.pack('<I', dt.itemsize) + struct.pack('<HH', 0, dt.itemsize * 8)
    if dt.kind == 'S':
        return bytes([0x13, 0x01, 0, 0]) + struct.pack('<I', max(dt.itemsize, 1))     # null-padded ASCII
    raise H5Error('unsupported dtype %r' % dt)


def _space_msg(shape):
    return struct.pack('<BBB5x', 1, len(shape), 0) + b''.join(struct.pack('<Q', int(d)) for d in shape)


def _message(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack('<HHB3x', mtype, len(data), flags) + data


def _attr_msg(name, value):
    if isinstance(value, (bytes, str)):
        if isinstance(value, str):
            value = value.encode('utf8')
        arr = np.array(value, dtype='S%d' % max(len(value), 1))
    else:
        arr = np.asarray(value)
        if arr.dtype.kind == 'U':
            arr = np.char.encode(arr, 'utf8')
        if arr.dtype.kind == 'O' or arr.size == 0 and arr.dtype.kind not in 'fSiu':
            arr = arr.astype('S1') if arr.size == 0 else np.array([bytes(x) for x in arr.ravel()])
    if arr.dtype.kind == 'S' and arr.dtype.itemsize == 0:
        arr = arr.astype('S1')
    nameb = name.encode('utf8') + b'\0'
    dt, sp = _dtype_msg(arr.dtype), _space_msg(arr.shape)
    body = struct.pack('<BBHHH', 1, 0, len(nameb), len(dt), len(sp)) + _pad8(nameb) + _pad8(dt) + _pad8(sp)
    body += np.ascontiguousarray(arr).tobytes()
    return _message(MSG_ATTRIBUTE, body)


def _object_header(messages):
    body = b''.join(messages)
    return struct.pack('<BBHII4x', 1, 0, len(messages), 1, len(body)) + body


def _write_dataset(out, arr):
    arr = np.ascontiguousarray(arr)
    if arr.dtype.byteorder == '>':
        arr = arr.astype(arr.dtype.newbyteorder('<'))
    raw = arr.tobytes()
    addr = out.alloc(raw) if raw else UNDEF
    msgs = [_message(MSG_DATASPACE, _space_msg(arr.shape)),
            _message(MSG_DATATYPE, _dtype_msg(arr.dtype), flags=1),
            _message(MSG_FILL, struct.pack('<BBBB', 2, 2, 0, 0)),
            _message(MSG_LAYOUT, struct.pack('<BBQQ', 3, 1, addr, len(raw)))]
    return out.alloc(_object_header(msgs))


CHUNK_K = 32     # default "indexed storage internal node K" of a version-0 superblock


def _write_gzip_dataset(out, arr, rows_per_chunk=None):
    import zlib
    arr = np.ascontiguousarray(arr)
    rank = arr.ndim
    if rank == 0 or arr.size == 0:
        return _write_dataset(out, arr)
    rpc = rows_per_chunk or max(1, min(arr.shape[0], (1 << 20) // max(1, arr[0].nbytes)))
    cdims = (rpc,) + arr.shape[1:]
    chunks = []
    for r0 in range(0, arr.shape[0], rpc):
        block = np.zeros(cdims, arr.dtype)
        part = arr[r0:r0 + rpc]
        block[:part.shape[0]] = part
        comp = zlib.compress(block.tobytes(), 4)
        chunks.append((r0, len(comp), out.alloc(comp)))
    if len(chunks) > 2 * CHUNK_K:
        raise H5Error('too many chunks for a single B-tree node')
    keysize = 8 + 8 * (rank + 1)
    node = b'TREE' + struct.pack('<BBHQQ', 1, 0, len(chunks), UNDEF, UNDEF)
    for r0, csize, caddr in chunks:
        node += struct.pack('<II', csize, 0) + struct.pack('<%dQ' % (rank + 1), *((r0,) + (0,) * rank))
        node += struct.pack('<Q', caddr)
    last = ((arr.shape[0] + rpc - 1) // rpc) * rpc
    node += struct.pack('<II', 0, 0) + struct.pack('<%dQ' % (rank + 1), *((last,) + (0,) * rank))
    node += b'\0' * (24 + 2 * CHUNK_K * 8 + (2 * CHUNK_K + 1) * keysize - len(node))
    bt = out.alloc(node)
    layout = struct.pack('<BBBQ', 3, 2, rank + 1, bt) + struct.pack('<%dI' % (rank + 1), *(cdims + (arr.dtype.itemsize,)))
    filt = struct.pack('<BB6x', 1, 1) + struct.pack('<HHHH', 1, 0, 1, 1) + struct.pack('<II', 4, 0)
    msgs = [_message(MSG_DATASPACE, _space_msg(arr.shape)),
            _message(MSG_DATATYPE, _dtype_msg(arr.dtype), flags=1),
            _message(MSG_FILL, struct.pack('<BBBB', 2, 3, 0, 0)),
            _message(0xB, filt),
            _message(MSG_LAYOUT, layout)]
    return out.alloc(_object_header(msgs))


def _write_group(out, g):
    """Returns (object header address, btree address, heap address)."""
    entries = []
    for name in sorted(g.children, key=lambda s: s.encode('utf8')):
        child = g.children[name]
        if isinstance(child, Group):
            addr, bt, hp = _write_group(out, child)
            entries.append((name, addr, 1, struct.pack('<QQ', bt, hp)))
        elif isinstance(child, GzipArray):
            entries.append((name, _write_gzip_dataset(out, child.arr), 0, b'\0' * 16))
        else:
            entries.append((name, _write_dataset(out, child), 0, b'\0' * 16))
    # local heap: offset 0 is the empty string
    heap = bytearray(8)
    offs = []
    for name, _, _, _ in entries:
        offs.append(len(heap))
        heap += _pad8(name.encode('utf8') + b'\0')
    heap_data_addr = out.alloc(bytes(heap))
    heap_addr = out.alloc(b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap), FREE_NULL, heap_data_addr))
    # symbol-table nodes of up to 8 entries, one B-tree leaf level
    per = 2 * LEAF_K
    chunks = [list(range(i, min(i + per, len(entries)))) for i in range(0, len(entries), per)]
    if len(chunks) > 2 * INTERNAL_K:
        raise H5Error('too many links in one group (%d)' % len(entries))
    snods = []
    for ch in chunks:
        body = b'SNOD' + struct.pack('<BBH', 1, 0, len(ch))
        for i in ch:
            name, addr, ctype, scratch = entries[i]
            body += struct.pack('<QQII', offs[i], addr, ctype, 0) + scratch
        body += b'\0' * (8 + per * 40 - len(body))
        snods.append(out.alloc(body))
    node = b'TREE' + struct.pack('<BBHQQ', 0, 0, len(chunks), UNDEF, UNDEF)
    keys = [0] + [offs[ch[-1]] for ch in chunks]
    for i, sn in enumerate(snods):
        node += struct.pack('<QQ', keys[i], sn)
    node += struct.pack('<Q', keys[len(snods)])
    node += b'\0' * (24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8 - len(node))
    bt_addr = out.alloc(node)
    msgs = [_message(MSG_STAB, struct.pack('<QQ', bt_addr, heap_addr))]
    for aname in g.attrs:
        msgs.append(_attr_msg(aname, g.attrs[aname]))
    return out.alloc(_object_header(msgs)), bt_addr, heap_addr


def write_file(path, root):
    out = _Out()
    root_addr, bt, hp = _write_group(out, root)
    while len(out.buf) % 8:
        out.buf.append(0)
    sb = SIGNATURE + struct.pack('<BBBBBBBB', 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack('<HHI', LEAF_K, INTERNAL_K, 0)
    sb += struct.pack('<QQQQ', 0, UNDEF, len(out.buf), UNDEF)
    sb += struct.pack('<QQII', 0, root_addr, 1, 0) + struct.pack('<QQ', bt, hp)
    assert len(sb) == 96
    out.buf[:96] = sb
    with open(path, 'wb') as f:
        f.write(bytes(out.buf))


# ------------------------------------------------------------------------------------------
# reader
# ------------------------------------------------------------------------------------------
class _In(object):
    def __init__(self, data):
        self.d = data

    def u(self, fmt, off):
        return struct.unpack_from('<' + fmt, self.d, off)


def _parse_dtype(b):
    cls, ver = b[0] & 0x0F, b[0] >> 4
    size = struct.unpack_from('<I', b, 4)[0]
    if cls == 1:
        order = '>' if b[1] & 1 else '<'
        return np.dtype(order + 'f%d' % size)
    if cls == 0:
        order = '>' if b[1] & 1 else '<'
        return np.dtype(order + ('i' if b[1] & 8 else 'u') + '%d' % size)
    if cls == 3:
        return np.dtype('S%d' % size)
    raise H5Error('unsupported datatype class %d (v%d)' % (cls, ver))


def _parse_space(b):
    ver = b[0]
    if ver == 1:
        rank = b[1]
        return tuple(struct.unpack_from('<%dQ' % rank, b, 8)) if rank else ()
    if ver == 2:
        rank, stype = b[1], b[3]
        if stype == 2:
            return (0,)
        return tuple(struct.unpack_from('<%dQ' % rank, b, 4)) if rank else ()
    raise H5Error('unsupported dataspace version %d' % ver)


def _read_messages(f, addr):
    ver = f.d[addr]
    if f.d[addr:addr + 4] == b'OHDR':
        raise H5Error('version-2 object headers are not supported')
    if ver != 1:
        raise H5Error('bad object header at %d' % addr)
    nmsg, _, size = f.u('HII', addr + 2)
    msgs, chunks = [], [(addr + 16, size)]
    while chunks and len(msgs) < nmsg:
        off, length = chunks.pop(0)
        end = off + length
        while off + 8 <= end and len(msgs) < nmsg:
            mtype, msize, flags = f.u('HHB', off)
            body = f.d[off + 8:off + 8 + msize]
            off += 8 + msize
            if mtype == MSG_CONT:
                chunks.append(struct.unpack_from('<QQ', body, 0))
            msgs.append((mtype, flags, body))
    return msgs


def _global_heap_object(f, addr, index):
    if f.d[addr:addr + 4] != b'GCOL':
        raise H5Error('bad global heap collection')
    size = f.u('Q', addr + 8)[0]
    off, end = addr + 16, addr + size
    while off + 16 <= end:
        idx, _, _, osize = f.u('HHIQ', off)
        if idx == 0:
            break
        if idx == index:
            return f.d[off + 16:off + 16 + osize]
        off += 16 + (osize + 7) // 8 * 8
    raise H5Error('global heap object %d not found' % index)


def _parse_attr(body, f=None):
    ver = body[0]
    if ver == 1:
        nlen, dlen, slen = struct.unpack_from('<HHH', body, 2)
        off = 8
        name = body[off:off + nlen].split(b'\0')[0].decode('utf8')
        off += (nlen + 7) // 8 * 8
        dt = body[off:off + dlen]
        off += (dlen + 7) // 8 * 8
        sp = body[off:off + slen]
        off += (slen + 7) // 8 * 8
    elif ver in (2, 3):
        nlen, dlen, slen = struct.unpack_from('<HHH', body, 2)
        off = 8 + (1 if ver == 3 else 0)
        name = body[off:off + nlen].split(b'\0')[0].decode('utf8')
        off += nlen
        dt = body[off:off + dlen]
        off += dlen
        sp = body[off:off + slen]
        off += slen
    else:
        raise H5Error('unsupported attribute version %d' % ver)
    shape = _parse_space(sp)
    if (dt[0] & 0x0F) == 9 and (dt[1] & 0x0F) == 1 and f is not None:
        # variable-length string(s) (h5py >= 3 stores bytes attributes this way): 16-byte
        # {length, global-heap collection address, object index} descriptors
        n = int(np.prod(shape)) if shape else 1
        vals = []
        for i in range(n):
            ln, gaddr, gidx = struct.unpack_from('<IQI', body, off + 16 * i)
            vals.append(bytes(_global_heap_object(f, gaddr, gidx)[:ln]) if ln else b'')
        if shape == ():
            return name, np.bytes_(vals[0])
        return name, np.array(vals, dtype='S%d' % max(1, max(len(v) for v in vals) if vals else 1)).reshape(shape)
    try:
        dtype = _parse_dtype(dt)
    except H5Error:
        return name, None
    n = int(np.prod(shape)) if shape else 1
    arr = np.frombuffer(body, dtype=dtype, count=n, offset=off).reshape(shape)
    if shape == ():
        return name, arr[()]
    return name, arr.copy()


def _read_group_entries(f, bt_addr, heap_addr):
    if f.d[heap_addr:heap_addr + 4] != b'HEAP':
        raise H5Error('bad local heap')
    heap_data = f.u('Q', heap_addr + 24)[0]
    out = []

    def name_at(off):
        s = heap_data + off
        e = f.d.find(b'\0', s)
        return f.d[s:e].decode('utf8')

    def walk(addr):
        if f.d[addr:addr + 4] != b'TREE':
            raise H5Error('bad B-tree node')
        ntype, level, used = f.u('BBH', addr + 4)
        if ntype != 0:
            raise H5Error('not a group B-tree')
        for i in range(used):
            child = f.u('Q', addr + 24 + 8 + i * 16)[0]
            if level > 0:
                walk(child)
            else:
                if f.d[child:child + 4] != b'SNOD':
                    raise H5Error('bad symbol table node')
                nsym = f.u('H', child + 6)[0]
                for k in range(nsym):
                    noff, oaddr = f.u('QQ', child + 8 + k * 40)
                    out.append((name_at(noff), oaddr))
    walk(bt_addr)
    return out


def _read_object(f, addr, lazy=False):
    msgs = _read_messages(f, addr)
    attrs = {}
    stab = shape = dtype = layout = filters = None
    for mtype, flags, body in msgs:
        if mtype == MSG_STAB:
            stab = struct.unpack_from('<QQ', body, 0)
        elif mtype == MSG_DATASPACE:
            shape = _parse_space(body)
        elif mtype == MSG_DATATYPE:
            dtype = _parse_dtype(body)
        elif mtype == MSG_LAYOUT:
            layout = body
        elif mtype == 0xB:
            filters = _parse_filters(body)
        elif mtype == MSG_ATTRIBUTE:
            k, v = _parse_attr(body, f)
            attrs[k] = v
        elif mtype in (0x2, 0x6):
            raise H5Error('new-style (link message) groups are not supported')
    if stab is not None:
        g = Group()
        g.attrs = attrs
        for name, oaddr in _read_group_entries(f, stab[0], stab[1]):
            g.children[name] = _read_object(f, oaddr, lazy)
        return g
    if layout is None or dtype is None or shape is None:
        raise H5Error('object at %d is neither a group nor a dataset' % addr)
    ds = Dataset(f, layout, tuple(shape), dtype, filters or [])
    return ds if lazy else ds.read()


class Dataset(object):
    """A dataset of an open file: shape / dtype from the object header alone; `read_rows(lo, hi)` touches
    (and, for chunked gzip data, inflates) only what rows [lo, hi) of the leading axis need."""

    def __init__(self, f, layout, shape, dtype, filters):
        self._f, self._layout, self._filters = f, layout, filters
        self.shape = shape
        self.dtype = dtype.newbyteorder('=')
        self._dt = dtype
        self._chunks = None
        ver, cls = layout[0], layout[1]
        if ver != 3 or cls not in (0, 1, 2):
            raise H5Error('unsupported data layout (version %d class %d)' % (ver, cls))
        self._cls = cls

    def __len__(self):
        return self.shape[0] if self.shape else 1

    def read(self):
        if not self.shape:
            return self._read_all()
        return self.read_rows(0, self.shape[0]).reshape(self.shape)

    def _read_all(self):                                   # scalar dataspace
        layout, f = self._layout, self._f
        if self._cls == 1:
            daddr = struct.unpack_from('<Q', layout, 2)[0]
            if daddr == UNDEF:
                return np.zeros((), self.dtype)
            return np.frombuffer(f.d, dtype=self._dt, count=1, offset=daddr).reshape(()).astype(self.dtype)
        if self._cls == 0:
            return np.frombuffer(layout, dtype=self._dt, count=1, offset=4).reshape(()).astype(self.dtype)
        raise H5Error('chunked scalar dataset')

    def read_rows(self, lo, hi):
        lo, hi = max(0, int(lo)), min(int(hi), self.shape[0])
        rest = self.shape[1:]
        per = int(np.prod(rest)) if rest else 1
        if hi <= lo:
            return np.zeros((0,) + rest, self.dtype)
        layout, f = self._layout, self._f
        if self._cls == 1:
            daddr = struct.unpack_from('<Q', layout, 2)[0]
            if daddr == UNDEF:
                return np.zeros((hi - lo,) + rest, self.dtype)
            a = np.frombuffer(f.d, dtype=self._dt, count=(hi - lo) * per, offset=daddr + lo * per * self._dt.itemsize)
            return a.reshape((hi - lo,) + rest).astype(self.dtype)
        if self._cls == 0:
            a = np.frombuffer(layout, dtype=self._dt, count=self.shape[0] * per, offset=4)
            return a.reshape(self.shape)[lo:hi].astype(self.dtype)
        return self._read_chunked_rows(lo, hi)

    def _chunk_table(self):
        """[(file offset, stored bytes, filter mask, chunk origin)] from the chunk B-tree, read once."""
        if self._chunks is not None:
            return self._chunks
        f, layout = self._f, self._layout
        ndim1 = layout[2]
        bt = struct.unpack_from('<Q', layout, 3)[0]
        self._cdims = struct.unpack_from('<%dI' % ndim1, layout, 11)[:-1]
        rank = ndim1 - 1
        keysize = 8 + 8 * ndim1
        table = []

        def walk(addr):
            if bytes(f.d[addr:addr + 4]) != b'TREE':
                raise H5Error('bad chunk B-tree node')
            ntype, level, used = f.u('BBH', addr + 4)
            if ntype != 1:
                raise H5Error('not a chunk B-tree')
            off = addr + 24
            for _ in range(used):
                csize, mask = f.u('II', off)
                coord = f.u('%dQ' % ndim1, off + 8)[:rank]
                child = f.u('Q', off + keysize)[0]
                off += keysize + 8
                if level > 0:
                    walk(child)
                else:
                    table.append((child, csize, mask, coord))

        if bt != UNDEF:
            walk(bt)
        self._chunks = table
        return table

    def _read_chunked_rows(self, lo, hi):
        import zlib
        f, dtype, filters, shape = self._f, self._dt, self._filters, self.shape
        table = self._chunk_table()
        cdims = self._cdims
        out = np.zeros((hi - lo,) + shape[1:], dtype=dtype)
        tasks = [t for t in table if t[3][0] < hi and t[3][0] + cdims[0] > lo]

        def decode(task):
            child, csize, mask, coord = task
            raw = bytes(f.d[child:child + csize])
            for k, (fid, cd) in reversed(list(enumerate(filters))):
                if mask & (1 << k):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    es = cd[0] if cd else dtype.itemsize
                    a = np.frombuffer(raw, np.uint8)
                    n = len(a) // es
                    raw = a[:n * es].reshape(es, n).T.tobytes() + a[n * es:].tobytes()
                else:
                    raise H5Error('unsupported HDF5 filter id %d' % fid)
            block = np.frombuffer(raw, dtype=dtype, count=int(np.prod(cdims))).reshape(cdims)
            # chunk rows [coord0, coord0 + cdims0) clipped to [lo, hi); the other axes clipped to the shape
            r0, r1 = max(coord[0], lo), min(coord[0] + cdims[0], hi)
            sl_out = (slice(r0 - lo, r1 - lo),) + tuple(slice(c, min(c + d, s)) for c, d, s in
                                                          zip(coord[1:], cdims[1:], shape[1:]))
            sl_in = (slice(r0 - coord[0], r1 - coord[0]),) + tuple(slice(0, so.stop - so.start) for so in sl_out[1:])
            out[sl_out] = block[sl_in]

        # chunks are independent and zlib.decompress / the copies release the GIL: inflate them on a small
        # thread pool (the batch blobs are ~250 KB per pair; one thread feeds ~300 pairs/s)
        pool = _decode_pool()
        if pool is None or len(tasks) < 2:
            for t in tasks:
                decode(t)
        else:
            list(pool.map(decode, tasks))
        return out.astype(self.dtype, copy=False)


def _parse_filters(body):
    ver, n = body[0], body[1]
    out = []
    off = 8 if ver == 1 else 2
    for _ in range(n):
        fid = struct.unpack_from('<H', body, off)[0]
        if ver == 1 or fid >= 256:
            nlen, flags, ncd = struct.unpack_from('<HHH', body, off + 2)
            off += 8
            off += (nlen + 7) // 8 * 8 if ver == 1 else nlen
        else:
            flags, ncd = struct.unpack_from('<HH', body, off + 2)
            off += 6
        cd = struct.unpack_from('<%dI' % ncd, body, off)
        off += 4 * ncd
        if ver == 1 and ncd % 2:
            off += 4
        out.append((fid, cd))
    return out


_POOL = None


def _decode_pool():
    global _POOL
    n = int(os.environ.get('L3_H5_THREADS', '0')) or min(16, os.cpu_count() or 1)
    if n <= 1:
        return None
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=n, thread_name_prefix='l3-h5')
    return _POOL


def _root(data, path):
    if bytes(data[:8]) != SIGNATURE:
        raise H5Error('not an HDF5 file: %s' % path)
    ver = data[8]
    if ver not in (0, 1):
        raise H5Error('superblock version %d is not supported (write the file with libver earliest)' % ver)
    if data[13] != 8 or data[14] != 8:
        raise H5Error('only 8-byte offsets/lengths are supported')
    base = 24 + (4 if ver == 1 else 0)
    f = _In(data)
    return f, f.u('Q', base + 32 + 8)[0]


def read_file(path):
    """Whole file -> Group tree with every dataset as an ndarray."""
    with open(path, 'rb') as fh:
        data = fh.read()
    f, root_addr = _root(data, path)
    return _read_object(f, root_addr)


class File(object):
    """Lazily opened file (memory-mapped): `File(path)[name]` is a `Dataset` whose rows are read --
    and, for the gzip-chunked batch blobs, inflated -- on demand.  Context manager."""

    def __init__(self, path):
        import mmap
        self.path = path
        self._fh = open(path, 'rb')
        try:
            self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:
            self._fh.close()
            raise H5Error('empty file: %s' % path)
        f, root_addr = _root(self._mm, path)
        self.root = _read_object(f, root_addr, lazy=True)

    def __getitem__(self, name):
        return self.root[name]

    def __contains__(self, name):
        try:
            self.root[name]
            return True
        except KeyError:
            return False

    def close(self):
        if self._mm is not None:
            try:
                self._mm.close()
            except BufferError:        # an ndarray view of the map is still alive; the GC closes it later
                pass
            self._mm = None
            self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
