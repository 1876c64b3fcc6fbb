"""Drop-in for the reference's opts.py (``from opts import parse_opts``)."""
from cstp_amd.opts import build_parser, parse_opts  # noqa: F401
