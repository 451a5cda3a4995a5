"""stand-in for astropy.table (golden generation only): a dict of equal-length columns with the row selections the
reference's prepare_sim uses (`t[mask]`, `t[a:b]`, `t[mask]['col']`, `t['new'] = column`, `len(t)` = rows)"""
import numpy as np


class Table(dict):
    def __init__(self, data=None, meta=None, copy=True):
        super().__init__(data or {})
        self.meta = meta or {}

    def __getitem__(self, key):
        if isinstance(key, str):
            return dict.__getitem__(self, key)
        if isinstance(key, (list, tuple)) and key and all(isinstance(k, str) for k in key):
            return Table({k: dict.__getitem__(self, k) for k in key}, meta=self.meta)
        return Table({k: np.asarray(v)[key] for k, v in dict.items(self)}, meta=self.meta)

    def __len__(self):
        cols = list(dict.values(self))
        if cols and all(hasattr(c, 'shape') and getattr(c, 'ndim', 0) >= 1 for c in cols) and len({len(c) for c in cols}) == 1:
            return len(cols[0])
        return dict.__len__(self)

    @property
    def colnames(self):
        return list(dict.keys(self))
