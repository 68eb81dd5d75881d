"""Test-only oracle package (see xattn_oracle.py). Never imported by the product path."""
