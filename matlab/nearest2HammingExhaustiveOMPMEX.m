function [idx2, d1, d2] = nearest2HammingExhaustiveOMPMEX(Abytes, Bbytes)
    %NEAREST2HAMMINGEXHAUSTIVEOMPMEX Replaces PP/mex/nearest2HammingExhaustiveOMPMEX.cpp (same outputs and tie rules).
    [idx2, d1, d2] = aps_mex('hamming_2nn', Abytes, Bbytes);
end
