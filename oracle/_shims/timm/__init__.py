"""Minimal stand-in for the 0.4.x-era timm API surface the reference imports."""
