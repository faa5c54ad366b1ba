"""MI355X-native SUCRe restoration engine (hot path of clementinboittiaux/sucre re-built for gfx950)."""
__version__ = '0.1.0'
