"""Stands where the reference's `pt_custom_ops` package is imported from."""
