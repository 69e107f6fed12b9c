"""Adapters that plug the MI355X kernels into third-party model code through its own extension points."""
