"""stand-in for astropy.table (golden generation only)"""


class Table(dict):
    def __init__(self, data=None, meta=None):
        super().__init__(data or {})
        self.meta = meta or {}
