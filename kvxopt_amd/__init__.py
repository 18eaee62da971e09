"""kvxopt_amd -- MI355X (gfx950) implementation of KVXOPT's cone-LP KKT factor/solve hot path.

Mirrors the reference's `kvxopt.cholmod` and (orthant part of) `kvxopt.misc` API on top of a
C-ABI HIP library (libkvxhip.so, include/kvxhip.h).  No CPU fallback.
"""
__version__ = "0.1.0"
