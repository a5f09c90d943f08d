export TMPDIR=/tmp; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_[A-Z_0-9a-z]*" | sort -u | tr '\n' ' '
