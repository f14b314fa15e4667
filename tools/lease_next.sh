#!/bin/bash
# what tools/lease_poller.sh submits next (edit to point at another lease)
exec bash tools/r6_first.sh
