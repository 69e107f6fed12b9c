"""Model adapters, mirroring the reference's touchnet/models/<name>/ layout."""
