function [idx2, d1, d2] = nearest2HammingExhaustiveMEX(Abytes, Bbytes)
    %NEAREST2HAMMINGEXHAUSTIVEMEX Replaces PP/mex/nearest2HammingExhaustiveMEX.cpp (same outputs and tie rules).
    [idx2, d1, d2] = aps_mex('hamming_2nn', Abytes, Bbytes);
end
