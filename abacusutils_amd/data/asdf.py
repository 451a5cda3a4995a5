"""Reader for the ASDF files of an Abacus simulation (halo_info, halo_rv / halo_pid, cleaned_halo_info, cleaned_rvpid, light-cone
catalogues) without the `asdf` package: the YAML tree is parsed with PyYAML, the binary blocks are located by their
`\\xd3BLK` headers and decoded here.  Abacus compresses blocks with its own 'blsc' scheme (abacusnbody/data/asdf.py:81-93,
128-181 of the reference): a block is a sequence of [4-byte big-endian length][one blosc frame]; the frames are decoded by
the C-Blosc library of the image (through the `blosc` package if there is one, else libblosc via ctypes).

`read_asdf(fn, fields=None)` -> (tree, {name: ndarray}) with the arrays of `tree['data']` (lazily: only `fields`)."""
import ctypes
import ctypes.util
import struct

import numpy as np

__all__ = ['read_asdf', 'AsdfFile']

_DT = {'float32': 'f4', 'float64': 'f8', 'int8': 'i1', 'int16': 'i2', 'int32': 'i4', 'int64': 'i8', 'uint8': 'u1',
       'uint16': 'u2', 'uint32': 'u4', 'uint64': 'u8', 'bool8': 'u1', 'complex64': 'c8', 'complex128': 'c16'}
_BLK = b'\xd3BLK'


class _Blosc:
    """blosc_decompress of the first C-Blosc found: the python package, or the shared library"""
    _fn = None

    @classmethod
    def get(cls):
        if cls._fn is not None:
            return cls._fn
        try:
            import blosc
            cls._fn = lambda frame: blosc.decompress(frame)
            return cls._fn
        except ImportError:
            pass
        names = [ctypes.util.find_library('blosc'), 'libblosc.so.1', 'libblosc.so', '/opt/conda/lib/libblosc.so.1',
                 '/usr/lib/x86_64-linux-gnu/libblosc.so.1']
        for nm in names:
            if not nm:
                continue
            try:
                lib = ctypes.CDLL(nm)
            except OSError:
                continue
            lib.blosc_cbuffer_sizes.argtypes = [ctypes.c_char_p] + [ctypes.POINTER(ctypes.c_size_t)] * 3
            lib.blosc_decompress.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t]
            lib.blosc_decompress.restype = ctypes.c_int

            def dec(frame, lib=lib):
                nbytes, cbytes, bs = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
                lib.blosc_cbuffer_sizes(frame, ctypes.byref(nbytes), ctypes.byref(cbytes), ctypes.byref(bs))
                buf = ctypes.create_string_buffer(max(nbytes.value, 1))
                r = lib.blosc_decompress(frame, buf, ctypes.c_size_t(nbytes.value))
                if r != nbytes.value:
                    raise OSError(f'blosc_decompress returned {r}, expected {nbytes.value}')
                return buf.raw[:nbytes.value]

            cls._fn = dec
            return cls._fn
        raise ImportError("Abacus ASDF blocks are blosc-compressed: neither the `blosc` package nor libblosc was found")


def _tree_loader():
    import yaml

    class Loader(yaml.SafeLoader):
        pass

    def any_tag(loader, suffix, node):        # !core/ndarray-1.0.0, !core/asdf-1.1.0, ...: plain containers
        if isinstance(node, yaml.MappingNode):
            return loader.construct_mapping(node, deep=True)
        if isinstance(node, yaml.SequenceNode):
            return loader.construct_sequence(node, deep=True)
        return loader.construct_scalar(node)

    Loader.add_multi_constructor('tag:', any_tag)      # tag:stsci.edu:asdf/..., tag:astropy.org:astropy/table/...
    Loader.add_multi_constructor('!', any_tag)
    return Loader


class AsdfFile:
    """an open Abacus ASDF file: `.tree` (dict; `tree['header']`, `tree['data']`), `.array(name)` decodes one column"""

    def __init__(self, fn):
        import yaml
        self.fn = str(fn)
        with open(fn, 'rb') as f:
            self._raw = f.read()
        p = self._raw.find(_BLK)
        head = self._raw if p < 0 else self._raw[:p]
        end = head.rfind(b'\n...')            # end of the YAML document
        text = (head if end < 0 else head[:end]).decode('utf-8', errors='replace')
        self.tree = yaml.load(text, Loader=_tree_loader())
        self._blocks = []                     # (compression, data offset, used, data size)
        while p >= 0 and self._raw[p:p + 4] == _BLK:
            (hsize,) = struct.unpack('>H', self._raw[p + 4:p + 6])
            flags, comp, alloc, used, dsize = struct.unpack('>I4sQQQ', self._raw[p + 6:p + 6 + 32])
            start = p + 6 + hsize
            self._blocks.append((comp, start, used, dsize))
            p = start + alloc
        self._cache = {}

    @property
    def header(self):
        return self.tree.get('header', {})

    def _columns(self, key):
        """{name: ndarray node} of tree[key]: a plain mapping of arrays (Abacus files), or an astropy table (the catalogues
        the reference's tests keep: `columns: [{data: ndarray, name: ...}]`)"""
        node = self.tree[key]
        if isinstance(node, dict) and isinstance(node.get('columns'), list):
            return {c['name']: c['data'] for c in node['columns']}
        return node

    def names(self, key='data'):
        return list(self._columns(key))

    def meta(self, key='data'):
        node = self.tree[key]
        return node.get('meta', {}) if isinstance(node, dict) else {}

    def _block(self, i):
        if i in self._cache:
            return self._cache[i]
        comp, start, used, dsize = self._blocks[i]
        if comp == b'blsc':
            dec = _Blosc.get()
            out, q, end = bytearray(), start, start + used
            while q < end:
                (n,) = struct.unpack('!I', self._raw[q:q + 4])
                out += dec(self._raw[q + 4:q + 4 + n])
                q += 4 + n
            if len(out) != dsize:
                raise OSError(f'{self.fn}: block {i} decodes to {len(out)} bytes, header says {dsize}')
            data = bytes(out)
        elif comp == b'\0\0\0\0':
            data = self._raw[start:start + dsize]
        else:
            raise NotImplementedError(f'{self.fn}: block compression {comp!r}')
        self._cache[i] = data
        return data

    def array(self, name, key='data'):
        node = self._columns(key)[name]
        if 'data' in node and 'source' not in node:     # a table column wraps its ndarray
            node = node['data']
        dt = np.dtype(_DT[node['datatype']]).newbyteorder('<' if node.get('byteorder', 'little') == 'little' else '>')
        shape = tuple(int(s) for s in node['shape'])
        if 'strides' in node:
            raise NotImplementedError(f'{self.fn}: strided view {name}')
        cnt = int(np.prod(shape, dtype=np.int64))
        a = np.frombuffer(self._block(int(node['source'])), dtype=dt, count=cnt, offset=int(node.get('offset', 0)))
        return a.reshape(shape).astype(dt.newbyteorder('='), copy=True)


def read_asdf(fn, fields=None, key='data'):
    af = AsdfFile(fn)
    names = af.names(key) if fields is None else list(fields)
    return af.tree, {n: af.array(n, key) for n in names}
