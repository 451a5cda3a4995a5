"""Bit-unpacking of Abacus particle data on the MI355X (mirror of abacusnbody.data.bitpacked)."""
