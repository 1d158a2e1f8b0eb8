// One instance of the frame decoder kernel (see dec_frames.h for why each has a file of its own).
#include "dec_frames.h"

DEC_FRAMES_LAUNCHER(fa_dec_frames_pair16, DEC_PAIR, 16, false)
