import sys
sys.path.insert(0,'.')
