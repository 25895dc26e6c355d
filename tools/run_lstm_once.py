"""Launch the persistent LSTM kernels a few times (for rocprofv3 counter runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import time_lstm_kernels as t
t.lstm_kernels(int(sys.argv[1]) if len(sys.argv) > 1 else 4096)
