/* empty: plain POSIX build */
