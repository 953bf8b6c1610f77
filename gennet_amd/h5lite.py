"""h5lite -- a dependency-free reader / writer for the HDF5 subset that Keras 2.x (through h5py 2.x, libver 'earliest') puts
into `.h5` / `.hdf5` model and weight files (bbhMahoGANy.py:1135-1142, :1171-1173, :1372-1375 save and load such files;
h5py is not part of this image).

Subset (HDF5 File Format Specification, version 1.x structures):
  * superblock version 0, 8-byte offsets and lengths;
  * version-1 object headers (+ continuation blocks), header messages: dataspace v1/v2, datatype (fixed-point, IEEE float,
    fixed-length string, variable-length string on read), fill value (skipped on read), data layout v3 contiguous / compact
    (chunked layouts raise), attribute v1/v2/v3, symbol table;
  * "old style" groups: symbol-table message -> version-1 B-tree of symbol-table nodes + local heap;
  * global heap collections (variable-length strings in attributes, read only).
Everything is little-endian.  The writer emits exactly these structures with libhdf5's default group parameters
(leaf K = 4, internal K = 16), so files it writes have the byte-level form of files h5py writes.

API:   f = File(path) ; f.attrs ; f['model_weights/dense_1/dense_1/kernel:0'].value ; f.keys() ; f.visit()
       w = Writer() ; g = w.root.create_group('model_weights') ; g.attrs['x'] = ... ; g.create_dataset('kernel:0', array) ; w.save(path)
"""
import os
import struct

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(IOError):
    pass


# =====================================================================================================================
# reader
# =====================================================================================================================

class _Datatype(object):
    """Parsed datatype message: numpy dtype (or 'vlen_str') + element size."""

    def __init__(self, buf, off=0):
        b0 = buf[off]
        self.cls = b0 & 0x0F
        self.version = b0 >> 4
        bits = buf[off + 1] | (buf[off + 2] << 8) | (buf[off + 3] << 16)
        self.size = struct.unpack_from('<I', buf, off + 4)[0]
        self.vlen_str = False
        if self.cls == 0:      # fixed point
            if bits & 1:
                raise H5Error('big-endian integers are not supported')
            self.dtype = np.dtype('<%s%d' % ('i' if bits & 0x08 else 'u', self.size))
        elif self.cls == 1:    # floating point
            if bits & 1:
                raise H5Error('big-endian floats are not supported')
            self.dtype = np.dtype('<f%d' % self.size)
        elif self.cls == 3:    # fixed-length string
            self.dtype = np.dtype('S%d' % self.size)
            self.charset = (bits >> 4) & 0x0F
        elif self.cls == 9:    # variable length
            if (bits & 0x0F) != 1:
                raise H5Error('variable-length sequences are not supported (only strings)')
            self.vlen_str = True
            self.dtype = None
        else:
            raise H5Error('datatype class %d is not supported' % self.cls)


def _dataspace(buf, off=0):
    """-> shape tuple (() for scalar, None for a null dataspace)."""
    version = buf[off]
    rank = buf[off + 1]
    flags = buf[off + 2]
    if version == 1:
        p = off + 8
    elif version == 2:
        if buf[off + 3] == 2:
            return None
        p = off + 4
    else:
        raise H5Error('dataspace message version %d is not supported' % version)
    return tuple(struct.unpack_from('<%dQ' % rank, buf, p)) if rank else ()


def _pad8(n):
    return (n + 7) & ~7


class Dataset(object):
    def __init__(self, f, name, msgs):
        self.file, self.name = f, name
        self.attrs = f._attrs(msgs)
        self._dt = self._shape = self._layout = None
        for t, body in msgs:
            if t == 0x01:
                self._shape = _dataspace(body)
            elif t == 0x03:
                self._dt = _Datatype(body)
            elif t == 0x08:
                self._layout = body
        if self._dt is None or self._shape is None or self._layout is None:
            raise H5Error('%s: incomplete dataset header' % name)

    @property
    def shape(self):
        return self._shape

    @property
    def dtype(self):
        return self._dt.dtype

    @property
    def value(self):
        body = self._layout
        if body[0] != 3:
            raise H5Error('%s: data layout message version %d is not supported' % (self.name, body[0]))
        n = int(np.prod(self._shape, dtype=np.int64)) if self._shape else 1
        nbytes = n * self._dt.size
        if body[1] == 1:       # contiguous
            addr, size = struct.unpack_from('<QQ', body, 2)
            if addr == UNDEF:
                raw = b'\x00' * nbytes
            else:
                raw = self.file._buf[self.file._base + addr: self.file._base + addr + nbytes]
        elif body[1] == 0:     # compact
            size = struct.unpack_from('<H', body, 2)[0]
            raw = bytes(body[4:4 + size])[:nbytes]
        else:
            raise H5Error('%s: chunked datasets are not supported' % self.name)
        if self._dt.vlen_str:
            raise H5Error('%s: variable-length datasets are not supported' % self.name)
        return np.frombuffer(raw, dtype=self._dt.dtype, count=n).reshape(self._shape).copy()

    def __getitem__(self, key):
        return self.value[key]


class Group(object):
    def __init__(self, f, name, msgs):
        self.file, self.name = f, name
        self.attrs = f._attrs(msgs)
        self._links = None
        self._stab = None
        for t, body in msgs:
            if t == 0x11:
                self._stab = struct.unpack_from('<QQ', body, 0)
        if self._stab is None:
            raise H5Error('%s: new-style (link message) groups are not supported' % name)

    def _load(self):
        if self._links is None:
            self._links = self.file._group_links(*self._stab)
        return self._links

    def keys(self):
        return list(self._load().keys())

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __iter__(self):
        return iter(self.keys())

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split('/') if p]:
            if not isinstance(node, Group):
                raise KeyError(path)
            links = node._load()
            if part not in links:
                raise KeyError(path)
            prefix = node.name.rstrip('/')
            node = node.file._object(links[part], prefix + '/' + part)
        return node

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def visit(self, fn=None, _prefix=''):
        """Depth-first walk in name order; returns the list of (path, object) when fn is None."""
        out = []
        for k in self.keys():
            o = self[k]
            path = _prefix + k
            out.append((path, o))
            if fn is not None:
                fn(path, o)
            if isinstance(o, Group):
                out.extend(o.visit(fn, path + '/'))
        return out


class File(Group):
    def __init__(self, path_or_bytes):
        if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
            self._buf = bytes(path_or_bytes)
        else:
            with open(path_or_bytes, 'rb') as fh:
                self._buf = fh.read()
        b = self._buf
        if b[:8] != SIGNATURE:
            raise H5Error('not an HDF5 file (no signature at offset 0)')
        ver = b[8]
        if ver not in (0, 1):
            raise H5Error('superblock version %d is not supported (only the version-0/1 layout h5py writes by default)' % ver)
        if b[13] != 8 or b[14] != 8:
            raise H5Error('only 8-byte offsets / lengths are supported')
        self.leaf_k, self.internal_k = struct.unpack_from('<HH', b, 16)
        p = 24 if ver == 0 else 28
        self._base, _fs, self._eof, _drv = struct.unpack_from('<QQQQ', b, p)
        p += 32
        _name_off, root_hdr, cache, _r = struct.unpack_from('<QQII', b, p)
        self._cache = {}
        Group.__init__(self, self, '/', self._messages(root_hdr))

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    # ---- low level ------------------------------------------------------------------------------------------------
    def _messages(self, addr):
        """[(type, body bytes)] of a version-1 object header, continuation blocks followed."""
        b = self._buf
        p = self._base + addr
        if b[p] != 1:
            raise H5Error('object header version %d at %d is not supported' % (b[p], addr))
        nmsg, _ref, hsize = struct.unpack_from('<HII', b, p + 2)
        blocks = [(p + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            q, size = blocks.pop(0)
            end = q + size
            while q + 8 <= end and len(out) < nmsg:
                t, sz, _flags = struct.unpack_from('<HHB', b, q)
                body = b[q + 8: q + 8 + sz]
                q += 8 + sz
                if t == 0x10:
                    off, ln = struct.unpack_from('<QQ', body, 0)
                    blocks.append((self._base + off, ln))
                out.append((t, body))
        return out

    def _object(self, addr, name):
        if addr not in self._cache:
            msgs = self._messages(addr)
            types = set(t for t, _ in msgs)
            self._cache[addr] = Dataset(self, name, msgs) if 0x08 in types else Group(self, name, msgs)
        return self._cache[addr]

    def _heap_string(self, heap_data, off):
        end = self._buf.index(b'\x00', heap_data + off)
        return self._buf[heap_data + off: end].decode('utf-8')

    def _group_links(self, btree, heap):
        b = self._buf
        hp = self._base + heap
        if b[hp:hp + 4] != b'HEAP':
            raise H5Error('bad local heap signature')
        _seg_size, _free, data_addr = struct.unpack_from('<QQQ', b, hp + 8)
        heap_data = self._base + data_addr
        links = {}

        def walk(addr):
            p = self._base + addr
            if b[p:p + 4] == b'TREE':
                ntype, level, used = struct.unpack_from('<BBH', b, p + 4)
                if ntype != 0:
                    raise H5Error('not a group B-tree')
                q = p + 24
                for i in range(used):
                    child = struct.unpack_from('<Q', b, q + 8 + 16 * i)[0]
                    walk(child)
            elif b[p:p + 4] == b'SNOD':
                n = struct.unpack_from('<H', b, p + 6)[0]
                for i in range(n):
                    name_off, hdr = struct.unpack_from('<QQ', b, p + 8 + 40 * i)
                    links[self._heap_string(heap_data, name_off)] = hdr
            else:
                raise H5Error('bad group node signature at %d' % addr)
        walk(btree)
        return dict(sorted(links.items()))

    def _global_heap_object(self, coll_addr, index):
        b = self._buf
        p = self._base + coll_addr
        if b[p:p + 4] != b'GCOL':
            raise H5Error('bad global heap signature')
        size = struct.unpack_from('<Q', b, p + 8)[0]
        q, end = p + 16, p + size
        while q + 16 <= end:
            idx, _ref, _r, osz = struct.unpack_from('<HHIQ', b, q)
            if idx == index:
                return b[q + 16: q + 16 + osz]
            if idx == 0:
                break
            q += 16 + _pad8(osz)
        raise H5Error('global heap object %d not found' % index)

    def _attr_value(self, dt, shape, raw):
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if dt.vlen_str:
            vals = []
            for i in range(n):
                ln, coll, idx = struct.unpack_from('<IQI', raw, 16 * i)
                vals.append(self._global_heap_object(coll, idx)[:ln].decode('utf-8') if ln else '')
            if shape == ():
                return vals[0]
            return np.array(vals, dtype=object).reshape(shape)
        arr = np.frombuffer(bytes(raw[:n * dt.size]), dtype=dt.dtype, count=n).reshape(shape if shape else ())
        if shape == ():
            v = arr[()]
            return bytes(v) if dt.cls == 3 else v
        return arr.copy()

    def _attrs(self, msgs):
        out = {}
        for t, body in msgs:
            if t != 0x0C:
                continue
            ver = body[0]
            name_sz, dt_sz, ds_sz = struct.unpack_from('<HHH', body, 2)
            if ver == 1:
                p = 8
                name = bytes(body[p:p + name_sz]).split(b'\x00')[0].decode('utf-8'); p += _pad8(name_sz)
                dt = _Datatype(body, p); p += _pad8(dt_sz)
                shape = _dataspace(body, p); p += _pad8(ds_sz)
            elif ver in (2, 3):
                if body[1] & 0x03:
                    raise H5Error('shared attribute datatypes / dataspaces are not supported')
                p = 8 if ver == 2 else 9
                name = bytes(body[p:p + name_sz]).split(b'\x00')[0].decode('utf-8'); p += name_sz
                dt = _Datatype(body, p); p += dt_sz
                shape = _dataspace(body, p); p += ds_sz
            else:
                raise H5Error('attribute message version %d is not supported' % ver)
            out[name] = None if shape is None else self._attr_value(dt, shape, body[p:])
        return out


# =====================================================================================================================
# writer
# =====================================================================================================================

def _dt_message(dtype):
    """Datatype message body for a numpy dtype (float32/64, (u)int8..64, fixed-length bytes)."""
    dtype = np.dtype(dtype)
    if dtype.kind == 'f' and dtype.itemsize in (4, 8):
        if dtype.itemsize == 4:
            return struct.pack('<BBBBIHHBBBBI', 0x11, 0x20, 31, 0, 4, 0, 32, 23, 8, 0, 23, 127)
        return struct.pack('<BBBBIHHBBBBI', 0x11, 0x20, 63, 0, 8, 0, 64, 52, 11, 0, 52, 1023)
    if dtype.kind in 'iu':
        return struct.pack('<BBBBIHH', 0x10, 0x08 if dtype.kind == 'i' else 0x00, 0, 0, dtype.itemsize, 0, 8 * dtype.itemsize)
    if dtype.kind == 'S':
        return struct.pack('<BBBBI', 0x13, 0x01, 0, 0, max(dtype.itemsize, 1))      # null-padded, ASCII (what h5py writes for numpy 'S')
    raise H5Error('dtype %s cannot be written' % dtype)


def _ds_message(shape, with_max=False):
    """Version-1 dataspace message body (rank 0 = scalar); datasets also carry max dims = dims, as libhdf5 writes them."""
    shape = tuple(int(s) for s in shape)
    flags = 1 if (with_max and shape) else 0
    body = struct.pack('<BBBBI', 1, len(shape), flags, 0, 0) + struct.pack('<%dQ' % len(shape), *shape)
    if flags:
        body += struct.pack('<%dQ' % len(shape), *shape)
    return body


def _as_array(value):
    """Python / numpy value -> little-endian numpy array with a writable dtype (str -> bytes, like Keras does itself)."""
    if isinstance(value, str):
        value = value.encode('utf-8')
    if isinstance(value, (list, tuple)) and value and isinstance(value[0], (str, bytes)):
        value = [v.encode('utf-8') if isinstance(v, str) else v for v in value]
    a = np.asarray(value)
    if a.dtype.kind == 'U':
        a = np.char.encode(a, 'utf-8')
    if a.dtype.kind == 'S' and a.dtype.itemsize == 0:
        a = a.astype('S1')
    if a.dtype.kind == 'b':
        a = a.astype(np.int8)
    if a.dtype.kind in 'fiu' and a.dtype.byteorder == '>':
        a = a.astype(a.dtype.newbyteorder('<'))
    if a.dtype.kind == 'f' and a.dtype.itemsize == 2:
        a = a.astype(np.float32)
    return a


def _message(mtype, body, flags=0):
    body = body + b'\x00' * (_pad8(len(body)) - len(body))
    return struct.pack('<HHBBBB', mtype, len(body), flags, 0, 0, 0) + body


def _attr_message(name, value):
    a = _as_array(value)
    nm = name.encode('utf-8') + b'\x00'
    dt = _dt_message(a.dtype)
    ds = _ds_message(a.shape)
    body = struct.pack('<BBHHH', 1, 0, len(nm), len(dt), len(ds))
    body += nm + b'\x00' * (_pad8(len(nm)) - len(nm))
    body += dt + b'\x00' * (_pad8(len(dt)) - len(dt))
    body += ds + b'\x00' * (_pad8(len(ds)) - len(ds))
    body += np.array(a, order='C').tobytes()
    if len(body) > 0xFFF8:
        raise H5Error('attribute %r is %d bytes: larger than one object-header message (64 KiB)' % (name, len(body)))
    return _message(0x0C, body)


class _WNode(object):
    def __init__(self):
        self.attrs = {}


class WDataset(_WNode):
    def __init__(self, data):
        _WNode.__init__(self)
        self.data = np.array(_as_array(data), order='C', copy=True)      # (ascontiguousarray would turn a 0-d value into (1,))


class WGroup(_WNode):
    def __init__(self):
        _WNode.__init__(self)
        self.children = {}

    def create_group(self, path):
        node = self
        for part in [p for p in path.split('/') if p]:
            nxt = node.children.get(part)
            if nxt is None:
                nxt = node.children[part] = WGroup()
            if not isinstance(nxt, WGroup):
                raise H5Error('%s is a dataset' % part)
            node = nxt
        return node

    def require_group(self, path):
        return self.create_group(path)

    def create_dataset(self, path, data):
        """`path` may contain '/', as Keras' weight names do ('dense_1/kernel:0'): intermediate groups are created."""
        parts = [p for p in path.split('/') if p]
        g = self.create_group('/'.join(parts[:-1])) if len(parts) > 1 else self
        if parts[-1] in g.children:
            raise H5Error('%s exists' % path)
        d = g.children[parts[-1]] = WDataset(data)
        return d


class Writer(object):
    LEAF_K, INTERNAL_K = 4, 16      # libhdf5 defaults, recorded in the superblock

    def __init__(self):
        self.root = WGroup()

    # ---- serialisation --------------------------------------------------------------------------------------------
    def tobytes(self):
        self._buf = bytearray(96)               # superblock v0 with the root symbol-table entry: 56 + 40 bytes
        root_hdr, root_btree, root_heap = self._write_group(self.root)
        eof = len(self._buf)
        sb = SIGNATURE + struct.pack('<BBBBBBBBHHI', 0, 0, 0, 0, 0, 8, 8, 0, self.LEAF_K, self.INTERNAL_K, 0)
        sb += struct.pack('<QQQQ', 0, UNDEF, eof, UNDEF)
        sb += struct.pack('<QQII', 0, root_hdr, 1, 0) + struct.pack('<QQ', root_btree, root_heap)
        assert len(sb) == 96
        self._buf[0:96] = sb
        return bytes(self._buf)

    def save(self, path):
        """Written to path + '.tmp' and moved into place (os.replace is atomic within a file system): a writer killed half-way never
        leaves a truncated file under the final name -- the previous checkpoint stays readable."""
        data = self.tobytes()
        tmp = path + '.tmp'
        try:
            with open(tmp, 'wb') as fh:
                fh.write(data)
            os.replace(tmp, path)
        except BaseException:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise

    def _alloc(self, blob):
        pad = _pad8(len(self._buf)) - len(self._buf)
        self._buf += b'\x00' * pad
        addr = len(self._buf)
        self._buf += blob
        return addr

    def _object_header(self, messages):
        body = b''.join(messages)
        hdr = struct.pack('<BBHII', 1, 0, len(messages), 1, len(body)) + b'\x00' * 4
        return self._alloc(hdr + body)

    def _write_dataset(self, d):
        a = d.data
        raw = a.tobytes()
        addr = self._alloc(raw) if raw else UNDEF
        # the four messages, versions and flags of a dataset header as h5py 2.x / libhdf5 1.8-1.10 write it (checked against
        # the real Keras files of the reference): dataspace v1 (+max dims), datatype (constant), fill value v2
        # {alloc late, write if-set, defined, size 0} (constant), layout v3 contiguous (constant)
        msgs = [_message(0x01, _ds_message(a.shape, with_max=True)),
                _message(0x03, _dt_message(a.dtype), flags=1),
                _message(0x05, struct.pack('<BBBBI', 2, 2, 2, 1, 0), flags=1),
                _message(0x08, struct.pack('<BBQQ', 3, 1, addr, len(raw)), flags=1)]
        msgs += [_attr_message(k, v) for k, v in d.attrs.items()]
        return self._object_header(msgs)

    def _write_group(self, g):
        # children first (their headers' addresses go into this group's symbol-table nodes)
        entries = []
        for name in sorted(g.children, key=lambda s: s.encode('utf-8')):
            child = g.children[name]
            if isinstance(child, WGroup):
                hdr, bt, hp = self._write_group(child)
                entries.append((name, hdr, 1, bt, hp))
            else:
                entries.append((name, self._write_dataset(child), 0, 0, 0))
        # local heap: the empty name at offset 0, then every link name, 8-byte aligned
        heap = bytearray(8)
        offs = []
        for name, _h, _c, _b, _p in entries:
            offs.append(len(heap))
            nm = name.encode('utf-8') + b'\x00'
            heap += nm + b'\x00' * (_pad8(len(nm)) - len(nm))
        free_off = len(heap)
        heap += struct.pack('<QQ', 1, 16)                      # one free block: next = 1 (none), size 16
        heap_data = self._alloc(bytes(heap))
        heap_addr = self._alloc(b'HEAP' + struct.pack('<BBBBQQQ', 0, 0, 0, 0, len(heap), free_off, heap_data))
        # symbol-table nodes of up to 2K entries, one B-tree level above them (more levels when > 2K_internal nodes)
        cap = 2 * self.LEAF_K
        leaves = []          # (address, heap offset of the largest name)
        for i in range(0, max(len(entries), 1), cap):
            chunk = entries[i:i + cap]
            body = b'SNOD' + struct.pack('<BBH', 1, 0, len(chunk))
            for j, (name, hdr, cache, bt, hp) in enumerate(chunk):
                body += struct.pack('<QQII', offs[i + j], hdr, cache, 0) + (struct.pack('<QQ', bt, hp) if cache == 1 else b'\x00' * 16)
            body += b'\x00' * (40 * (cap - len(chunk)))
            leaves.append((self._alloc(body), offs[i + len(chunk) - 1] if chunk else 0))
        level = 0
        nodes = leaves
        while True:
            fan = 2 * self.INTERNAL_K
            parents = []
            for i in range(0, len(nodes), fan):
                grp = nodes[i:i + fan]
                parents.append((grp, grp[-1][1]))
            written = []
            for idx, (grp, last_key) in enumerate(parents):
                body = b'TREE' + struct.pack('<BBH', 0, level, len(grp))
                body += struct.pack('<QQ', UNDEF, UNDEF)       # siblings patched below
                key = 0 if idx == 0 else parents[idx - 1][1]
                body += struct.pack('<Q', key)
                for addr, k in grp:
                    body += struct.pack('<QQ', addr, k)
                body += b'\x00' * (16 * (fan - len(grp)))
                written.append((self._alloc(body), last_key))
            for idx, (addr, _k) in enumerate(written):         # sibling pointers
                left = written[idx - 1][0] if idx > 0 else UNDEF
                right = written[idx + 1][0] if idx + 1 < len(written) else UNDEF
                self._buf[addr + 8: addr + 24] = struct.pack('<QQ', left, right)
            if len(written) == 1:
                btree = written[0][0]
                break
            nodes = written
            level += 1
        msgs = [_message(0x11, struct.pack('<QQ', btree, heap_addr))]
        msgs += [_attr_message(k, v) for k, v in g.attrs.items()]
        return self._object_header(msgs), btree, heap_addr


def dump_structure(path_or_bytes):
    """{'attrs': {path: {name: summary}}, 'datasets': {path: [shape, dtype, crc]}} -- what the golden fixture records."""
    import zlib
    f = File(path_or_bytes)

    def summ(v):
        if isinstance(v, bytes):
            return {'bytes_len': len(v), 'crc': zlib.crc32(v) & 0xFFFFFFFF}
        if isinstance(v, str):
            return {'str_len': len(v), 'crc': zlib.crc32(v.encode('utf-8')) & 0xFFFFFFFF}
        a = np.asarray(v)
        if a.dtype.kind == 'S':
            return {'strings': [s.decode('utf-8') for s in a.ravel().tolist()]}
        if a.dtype == object:
            return {'strings': [str(s) for s in a.ravel().tolist()]}
        return {'shape': list(a.shape), 'dtype': str(a.dtype), 'crc': zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF}

    out = {'attrs': {'/': {k: summ(v) for k, v in f.attrs.items()}}, 'groups': [], 'datasets': {}}
    for path, o in f.visit():
        if o.attrs:
            out['attrs'][path] = {k: summ(v) for k, v in o.attrs.items()}
        if isinstance(o, Dataset):
            v = o.value
            out['datasets'][path] = [list(o.shape), str(o.dtype), zlib.crc32(np.ascontiguousarray(v).tobytes()) & 0xFFFFFFFF]
        else:
            out['groups'].append(path)
    return out
